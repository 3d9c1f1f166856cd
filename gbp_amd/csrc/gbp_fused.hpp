// gbp_fused.hpp -- the whole synchronous_iteration (gbp.py:86-92) except the camera solve, as ONE kernel.
//
// Why: the general sweep re-reads every message after the barrier (variable stage) and gathers the
// camera messages through an index list (8-byte gathers = 8x read amplification).  Here each
// message is read once and written once per sweep, and everything else stays on chip:
//
//   * factors are stored landmark-major, so a TILE of <= 256 consecutive factors owns a contiguous
//     run of landmarks completely: the new landmark messages go through LDS and the landmark
//     beliefs (prior + sum in adj_factors order, gbp.py:182-188) are formed in the same kernel;
//   * camera messages are accumulated into a per-workgroup LDS table acc[C][27] (500 cameras =
//     108 KB of the CU's 160 KB LDS).  Lanes of a tile that hit the same camera are serialised by a
//     pre-computed rank kept in the state word (round r: lanes with rank r add, then a barrier) -> no atomics, bitwise
//     reproducible.  One workgroup per CU walks a fixed contiguous range of tiles, then stores its
//     table; k_cam_reduce_blocks sums the per-workgroup tables in workgroup order.
//   * one wave per SIMD (the LDS table allows one workgroup per CU) with the 512-VGPR budget that
//     brings: the NEXT tile's streaming inputs are loaded into a second register set before the
//     current tile is computed, so HBM latency overlaps the fp64 maths.
//
// HBM traffic per sweep: F*(47 read + 36 written doubles + 10 B of indices) + L*33 doubles + the
// workgroup tables (256 * C * 27 doubles written and read once) -- below the "algorithmic" 1072 B per
// factor of SURVEY.md 8d, which assumed a second pass over the messages.
//
// Landmarks with more than 256 factors do not fit a tile: their factors form tiles with nl = 0 (messages
// only, belief read from HBM) and their beliefs are formed afterwards by k_lmk_belief_list.
// If C*27 doubles + tile buffers exceed the LDS, the plan stays disabled and the general sweep runs.
#pragma once
#include "gbp_kernels.hpp"
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace gbp {

constexpr int MAX_TILE = 512;      // widest tile (= workgroup size) the fused sweep is built for
constexpr int TILE_LMKS = 128;     // landmarks staged per tile (LDS rows)
constexpr int LDS_BYTES = 160 * 1024;

struct FusedArgs {
    const unsigned *meta;       // [Fp] per factor: camera index | (landmark slot inside its tile) << 24
    double *block_partials;     // [n_blocks][C*27]
    int acc_doubles;            // C*27
    int dbg;                    // experiment switches (GBP_FUSED_DBG): 1 no ticket wait, 2 no accumulation, 4 no landmark phase
};
constexpr int META_CAM_BITS = 24;

struct Stream {                 // everything a factor streams from HBM each sweep
    double x0[9], z[2], eC[6], MC[21], eL[3], ML[6], avar;
    int st;
    unsigned meta;
};

template <int LOSS>
GBP_DEV void load_stream(const Params &p, const FusedArgs &a, int f, Stream &s)
{
    const size_t Fp = (size_t)p.Fp;
    s.meta = a.meta[f];
    s.st = p.state[f];
#pragma unroll
    for (int k = 0; k < 9; ++k) s.x0[k] = p.x0[k * Fp + f];
    s.z[0] = p.z[f]; s.z[1] = p.z[Fp + f];
#pragma unroll
    for (int k = 0; k < 6; ++k) s.eC[k] = p.mc[k * Fp + f];
#pragma unroll
    for (int k = 0; k < 21; ++k) s.MC[k] = p.mc[(6 + k) * Fp + f];
#pragma unroll
    for (int k = 0; k < 3; ++k) s.eL[k] = p.ml[k * Fp + f];
#pragma unroll
    for (int k = 0; k < 6; ++k) s.ML[k] = p.ml[(3 + k) * Fp + f];
    s.avar = (LOSS != 0) ? p.avar[f] : p.sigma2;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a workgroup-scope release
// of GLOBAL stores, which on gfx950 is s_waitcnt vmcnt(0): it would drain the next tile's prefetch (loads
// and stores share vmcnt) at every barrier.  No lane ever reads another lane's global stores inside this
// kernel, so only lgkmcnt (LDS) has to be waited for.
GBP_DEV void lds_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0), vmcnt/expcnt untouched
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

GBP_DEV void load_lmk_lane(const Params &p, int l, double (&lb)[21], int &row0, int &row1)
{
    const size_t Lp = (size_t)p.Lp;
    row0 = p.lptr[l]; row1 = p.lptr[l + 1];
#pragma unroll
    for (int k = 0; k < 9; ++k) lb[k] = p.lbel[k * Lp + l];
#pragma unroll
    for (int k = 0; k < 3; ++k) lb[9 + k] = p.lmu[k * Lp + l];
#pragma unroll
    for (int k = 0; k < 9; ++k) lb[12 + k] = p.lprior[k * Lp + l];
}

template <int LOSS, bool PREFETCH, int TILE>
__global__ __launch_bounds__(TILE, TILE / 256) void k_sweep_fused(Params p, FusedArgs a, const int4 *__restrict__ tiles,
                                                         const int *__restrict__ blk_begin)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *acc = smem;                                   // [C][27] camera accumulators
    double *s_ml = smem + ((a.acc_doubles + 1) & ~1);     // [TILE][9] new landmark messages of the tile
    double *s_lb = s_ml + TILE * 9;                       // [TILE_LMKS][12] landmark belief (9) + mean (3) of the tile
    const int tid = threadIdx.x;
    const size_t Fp = (size_t)p.Fp, Lp = (size_t)p.Lp;
    for (int i = tid; i < a.acc_doubles; i += TILE) acc[i] = 0.0;

    const int tb = blk_begin[blockIdx.x], te = blk_begin[blockIdx.x + 1];   // uniform: scalar loads
    Stream cur, nxt;
    double lbn[21];                                       // landmark lane: belief 9 | mean 3 | prior 9 of the tile to come
    int rown0 = 0, rown1 = 0;                             //                its factor range [lptr[l], lptr[l+1])
    int4 td = make_int4(0, 0, 0, 0), tdn = make_int4(0, 0, 0, 0);
    if (tb < te) {
        td = tiles[tb];
        if (tid < (td.z & 0xffff)) load_stream<LOSS>(p, a, td.x + tid, cur);
        if (tid < max(td.z >> 16, 1)) load_lmk_lane(p, td.y + tid, lbn, rown0, rown1);
    }

    for (int t = tb; t < te; ++t) {
        const int f0 = td.x, l0 = td.y, nf = td.z & 0xffff, nl = td.z >> 16, maxrank = td.w;
        const bool active = tid < nf;
        // (1) issue the camera gather for THIS tile first: vmcnt retires in order, so anything issued
        //     before it (and nothing after it) has to land before the maths can start
        double etaC[6], lamC[21], muC[6];
        const int cam = (int)(cur.meta & ((1u << META_CAM_BITS) - 1u));
        if (active) load_cam_record(p.cbel + (size_t)cam * CAMREC, etaC, lamC, muC);
        // (2) stage this tile's landmark beliefs for its factor lanes; keep the prior for step (5)
        double pri[9];
        const int row0 = rown0 - f0, row1 = rown1 - f0;
        if (tid < max(nl, 1)) {                           // chunk tiles (nl == 0): lane 0 stages the over-sized landmark
#pragma unroll
            for (int k = 0; k < 12; ++k) s_lb[tid * 12 + k] = lbn[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) pri[k] = lbn[12 + k];
        }
        // (3) stream the NEXT tile into the second register set while this one is computed
        if (PREFETCH && t + 1 < te) {
            tdn = tiles[t + 1];
            if (tid < (tdn.z & 0xffff)) load_stream<LOSS>(p, a, tdn.x + tid, nxt);
            if (tid < max(tdn.z >> 16, 1)) load_lmk_lane(p, tdn.y + tid, lbn, rown0, rown1);
        }
        lds_barrier();

        // (4) per-factor maths, message stores, new landmark messages to LDS
        if (active) {
            const int f = f0 + tid;
            double etaL[3], lamL[6], muL[3];
            const double *lb = s_lb + (cur.meta >> META_CAM_BITS) * 12;
#pragma unroll
            for (int k = 0; k < 3; ++k) etaL[k] = lb[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) lamL[k] = lb[3 + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) muL[k] = lb[9 + k];
            bool relin;
            factor_step<LOSS>(p, cur.x0, cur.z, cur.st, cur.avar, etaC, lamC, muC, etaL, lamL, muL,
                              cur.eC, cur.MC, cur.eL, cur.ML, relin);
            if (relin) {
#pragma unroll
                for (int k = 0; k < 9; ++k) p.x0[k * Fp + f] = cur.x0[k];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) p.mc[k * Fp + f] = cur.eC[k];
#pragma unroll
            for (int k = 0; k < 21; ++k) p.mc[(6 + k) * Fp + f] = cur.MC[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) { p.ml[k * Fp + f] = cur.eL[k]; s_ml[tid * 9 + k] = cur.eL[k]; }
#pragma unroll
            for (int k = 0; k < 6; ++k) { p.ml[(3 + k) * Fp + f] = cur.ML[k]; s_ml[tid * 9 + 3 + k] = cur.ML[k]; }
            p.state[f] = cur.st;
            if (LOSS != 0) p.avar[f] = cur.avar;
        }
        lds_barrier();

        // (5) landmark beliefs of the tile: prior + messages in adj_factors order (gbp.py:182-193)
        if (tid < nl) {
            const int l = l0 + tid;
            double b[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) b[k] = pri[k];
            for (int r = row0; r < row1; ++r) {
#pragma unroll
                for (int k = 0; k < 9; ++k) b[k] += s_ml[r * 9 + k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) p.lbel[k * Lp + l] = b[k];
            double eta[3] = {b[0], b[1], b[2]}, lam[6] = {b[3], b[4], b[5], b[6], b[7], b[8]}, mu[3];
            spd_solve<3>(lam, eta, mu);
#pragma unroll
            for (int k = 0; k < 3; ++k) p.lmu[k * Lp + l] = mu[k];
        }

        // (6) camera accumulation, same-camera lanes serialised by rank
        const int rank = state_rank(cur.st);
        for (int r = 0; r <= maxrank; ++r) {
            if (active && rank == r) {
                double *dst = acc + cam * 27;
#pragma unroll
                for (int k = 0; k < 6; ++k) dst[k] += cur.eC[k];
#pragma unroll
                for (int k = 0; k < 21; ++k) dst[6 + k] += cur.MC[k];
            }
            lds_barrier();
        }

        if (t + 1 < te) {
            if (PREFETCH) {
                cur = nxt;
                td = tdn;
            } else {
                td = tiles[t + 1];
                if (tid < (td.z & 0xffff)) load_stream<LOSS>(p, a, td.x + tid, cur);
                if (tid < max(td.z >> 16, 1)) load_lmk_lane(p, td.y + tid, lbn, rown0, rown1);
            }
        }
    }
    double *out = a.block_partials + (size_t)blockIdx.x * a.acc_doubles;
    for (int i = tid; i < a.acc_doubles; i += TILE) out[i] = acc[i];
}

// ---------------------------------------------------------------------------------------------
// Wave-autonomous variant: a TILE is 64 consecutive factors (one wavefront) that own <= 32 whole
// landmarks.  The 8 waves of a workgroup pull tiles from the workgroup's fixed range through an LDS
// counter and never meet at an s_barrier: while one wave waits for HBM another does fp64 maths on
// the same SIMD.  Determinism is kept by a TICKET: a wave may add its tile's camera messages to the
// shared table acc[C][27] only when all earlier tiles of the workgroup have done so (LDS word
// `done`), and lanes of the tile that hit the same camera add in rank order.  So the summation order
// is (tile index, rank) -- independent of which wave ran which tile and of timing.
constexpr int WTILE = 64;
constexpr int WAT_WAVES = 8;
constexpr int WTILE_LMKS = 32;
constexpr int WAVE_LDS_DOUBLES = WTILE * 9;        // per-wave scratch: [32][12] staged beliefs, then [64][9] messages

// Identity on `v` that the compiler must evaluate after `dep` exists: chains a load's address to a
// value so the load cannot be scheduled earlier (register-pressure control, see k_sweep_wat).
template <typename T>
GBP_DEV T after(T v, double dep)
{
    asm volatile("" : "+v"(v) : "v"(dep));
    return v;
}

GBP_DEV void wave_lds_sync()
{
    // all LDS traffic of this wave issued so far has completed; nothing may be moved across
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int LOSS, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void k_sweep_wat(Params p, FusedArgs a, const int4 *__restrict__ tiles,
                                                                          const int *__restrict__ blk_begin)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *acc = smem;                                              // [C][27]
    double *wl = smem + ((a.acc_doubles + 1) & ~1) + (threadIdx.x >> 6) * WAVE_LDS_DOUBLES;
    int *ctl = reinterpret_cast<int *>(smem + ((a.acc_doubles + 1) & ~1) + NWAVES * WAVE_LDS_DOUBLES);   // {next, done}
    const int tid = threadIdx.x, lane = tid & 63;
    const size_t Fp = (size_t)p.Fp, Lp = (size_t)p.Lp;
    for (int i = tid; i < a.acc_doubles; i += NWAVES * 64) acc[i] = 0.0;
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; }
    __syncthreads();
    const int tb = blk_begin[blockIdx.x], ntl = blk_begin[blockIdx.x + 1] - tb;

    for (;;) {
        int ti = 0;
        if (lane == 0) ti = atomicAdd(&ctl[0], 1);
        ti = __builtin_amdgcn_readfirstlane(ti);
        if (ti >= ntl) break;
        const int4 td = tiles[tb + ti];
        const int f0 = td.x, l0 = td.y, nf = td.z & 0xffff, nl = td.z >> 16, maxrank = td.w;
        const bool active = lane < nf;
        const int f = f0 + lane;

        // landmark lanes: belief 9 | mean 3 | prior 9 and the factor range of "their" landmark
        double lb[21];
        int row0 = 0, row1 = 0;
        if (lane < max(nl, 1)) load_lmk_lane(p, l0 + lane, lb, row0, row1);
        // factor lanes, first wave of loads: what the linearisation needs
        unsigned meta = 0;
        int st = 0;
        double x0[9], z[2], avar = p.sigma2;
        double eC[6], eLo[3], MLo[6];                      // old messages: everything except the 21 doubles of M_C
        if (active) {
            meta = a.meta[f];
            st = p.state[f];
#pragma unroll
            for (int k = 0; k < 9; ++k) x0[k] = p.x0[k * Fp + f];
            z[0] = p.z[f]; z[1] = p.z[Fp + f];
            if (LOSS != 0) avar = p.avar[f];
#pragma unroll
            for (int k = 0; k < 6; ++k) eC[k] = p.mc[k * Fp + f];
#pragma unroll
            for (int k = 0; k < 3; ++k) eLo[k] = p.ml[k * Fp + f];
#pragma unroll
            for (int k = 0; k < 6; ++k) MLo[k] = p.ml[(3 + k) * Fp + f];
        }
        const int cam = (int)(meta & ((1u << META_CAM_BITS) - 1u));
        const double2 *crec = reinterpret_cast<const double2 *>(p.cbel + (size_t)((a.dbg & 8) ? 0 : cam) * CAMREC);
        double muC[6];
        if (active) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { const double2 v = crec[i]; muC[2 * i] = v.x; muC[2 * i + 1] = v.y; }
        }
        // stage the tile's landmark beliefs / means through the wave's LDS scratch
        if (lane < max(nl, 1)) {
#pragma unroll
            for (int k = 0; k < 12; ++k) wl[lane * 12 + k] = lb[k];
        }
        wave_lds_sync();
        double etaL[3], lamL[6], muL[3];
        if (active) {
            const double *src = wl + (meta >> META_CAM_BITS) * 12;
#pragma unroll
            for (int k = 0; k < 3; ++k) etaL[k] = src[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) lamL[k] = src[3 + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) muL[k] = src[9 + k];
        }
        wave_lds_sync();                                   // scratch is free again

        double MC[21], eLn[3], MLn[6];
        if (active) {
            Lin L;
            const bool relin = factor_prepare<LOSS>(p, x0, z, st, avar, muC, muL, L);
            if (relin) {
#pragma unroll
                for (int k = 0; k < 9; ++k) p.x0[k * Fp + f] = x0[k];
            }
            p.state[f] = st;
            if (LOSS != 0) p.avar[f] = avar;
            // Second (and last) round trip: the camera belief (L2) and the old M_C (HBM).  Their addresses are
            // chained to the linearisation result (after()), so the compiler cannot hoist these 48 doubles into
            // the linearisation's live range: with two waves per SIMD a wave has 256 registers, and the other
            // wave covers the exposed latency.
            {
                double etaC[6], lamC[21];
                const double2 *c2 = after(crec, L.rho[0]);
                const int f1 = after(f, L.rho[1]);
#pragma unroll
                for (int i = 0; i < 3; ++i) { const double2 v = c2[3 + i]; etaC[2 * i] = v.x; etaC[2 * i + 1] = v.y; }
#pragma unroll
                for (int i = 0; i < 10; ++i) { const double2 v = c2[6 + i]; lamC[2 * i] = v.x; lamC[2 * i + 1] = v.y; }
                lamC[20] = c2[16].x;
#pragma unroll
                for (int k = 0; k < 21; ++k) lamC[k] -= p.mc[(6 + k) * Fp + f1];   // cavity Lambda
#pragma unroll
                for (int k = 0; k < 6; ++k) etaC[k] -= eC[k];                     // cavity eta
                message_to_landmark_cavity(L, etaC, lamC, eLo, eLn, MLn);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { p.ml[k * Fp + f] = eLn[k]; wl[lane * 9 + k] = eLn[k]; }
#pragma unroll
            for (int k = 0; k < 6; ++k) { p.ml[(3 + k) * Fp + f] = MLn[k]; wl[lane * 9 + 3 + k] = MLn[k]; }
#pragma unroll
            for (int k = 0; k < 6; ++k) lamL[k] -= MLo[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) etaL[k] -= eLo[k];
            message_to_camera_cavity(L, etaL, lamL, eC, MC);
#pragma unroll
            for (int k = 0; k < 6; ++k) p.mc[k * Fp + f] = eC[k];
#pragma unroll
            for (int k = 0; k < 21; ++k) p.mc[(6 + k) * Fp + f] = MC[k];
        }
        wave_lds_sync();

        // landmark beliefs of the tile: prior + messages in adj_factors order (gbp.py:182-193)
        if (lane < nl && !(a.dbg & 4)) {
            const int l = l0 + lane;
            double b[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) b[k] = lb[12 + k];
            for (int r = row0 - f0; r < row1 - f0; ++r) {
#pragma unroll
                for (int k = 0; k < 9; ++k) b[k] += wl[r * 9 + k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) p.lbel[k * Lp + l] = b[k];
            double eta[3] = {b[0], b[1], b[2]}, lam[6] = {b[3], b[4], b[5], b[6], b[7], b[8]}, mu[3];
            spd_solve<3>(lam, eta, mu);
#pragma unroll
            for (int k = 0; k < 3; ++k) p.lmu[k * Lp + l] = mu[k];
        }

        // ticket: camera accumulation strictly in tile order
        if (!(a.dbg & 1)) while (__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ti) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
        const int rank = state_rank(st);
        for (int r = 0; r <= maxrank; ++r) {
            if (active && rank == r && !(a.dbg & 2)) {     // one lane per camera in a round: ds_add_f64 is a plain RMW here
                double *dst = acc + cam * 27;
#pragma unroll
                for (int k = 0; k < 6; ++k) unsafeAtomicAdd(dst + k, eC[k]);
#pragma unroll
                for (int k = 0; k < 21; ++k) unsafeAtomicAdd(dst + 6 + k, MC[k]);
            }
        }
        wave_lds_sync();
        if (lane == 0) __hip_atomic_store(&ctl[1], ti + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    double *out = a.block_partials + (size_t)blockIdx.x * a.acc_doubles;
    for (int i = tid; i < a.acc_doubles; i += NWAVES * 64) out[i] = acc[i];
}

// partial[e] = sum over workgroups (fixed order) of block_partials[b][e]
__global__ __launch_bounds__(BLOCK) void k_cam_reduce_blocks(const double *__restrict__ block_partials, int n_blocks,
                                                             int n, double *__restrict__ partial)
{
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    double s = 0.0;
    int b = 0;
    for (; b + 8 <= n_blocks; b += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = block_partials[(size_t)(b + j) * n + e];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; b < n_blocks; ++b) s += block_partials[(size_t)b * n + e];
    partial[e] = s;
}

// beliefs of the landmarks that are larger than a tile
__global__ __launch_bounds__(64) void k_lmk_belief_list(Params p, const int *__restrict__ list, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int l = list[i];
    const size_t Fp = (size_t)p.Fp, Lp = (size_t)p.Lp;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = p.lprior[k * Lp + l];
    const int f1 = p.lptr[l + 1];
    for (int f = p.lptr[l]; f < f1; ++f) {
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += p.ml[k * Fp + f];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) p.lbel[k * Lp + l] = acc[k];
    double eta[3] = {acc[0], acc[1], acc[2]}, lam[6] = {acc[3], acc[4], acc[5], acc[6], acc[7], acc[8]}, mu[3];
    spd_solve<3>(lam, eta, mu);
#pragma unroll
    for (int k = 0; k < 3; ++k) p.lmu[k * Lp + l] = mu[k];
}

// ------------------------------------------------------------------------------------ host --

struct FusedPlan {
    bool enabled = false;
    bool prefetch = true;
    bool wat = false;
    int tile = 256;
    int n_tiles = 0, n_blocks = 0, n_big = 0;
    size_t shmem = 0;
    FusedArgs args{};
    const int4 *d_tiles = nullptr;
    const int *d_blk = nullptr;
    int *d_big = nullptr;
    std::vector<void *> allocs;
};

inline void fused_destroy(FusedPlan &pl)
{
    for (void *q : pl.allocs) (void)hipFree(q);
    pl.allocs.clear();
    pl.enabled = false;
}

template <typename T>
inline int fused_upload(FusedPlan &pl, T **dst, const T *src, size_t n, hipStream_t stream)
{
    void *q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return -1;
    pl.allocs.push_back(q);
    if (src && n) {
        if (hipMemcpyAsync(q, src, n * sizeof(T), hipMemcpyHostToDevice, stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(stream) != hipSuccess) return -1;
    }
    *dst = static_cast<T *>(q);
    return 0;
}

// Build tiles / ranks / workgroup ranges from the landmark CSR (internal order) and per-factor cameras.
inline int fused_plan(FusedPlan &pl, const Params &p, const std::vector<int32_t> &lptr, const std::vector<int32_t> &fcam,
                      std::vector<int32_t> &state, hipStream_t stream, int n_cus)
{
    const int acc_doubles = p.C * 27;
    const char *env_mode = getenv("GBP_FUSED_MODE");                          // wat (default) | block256 | block512
    pl.wat = !(env_mode && env_mode[0] == 'b');
    const int TILE = pl.wat ? WTILE : ((env_mode && atoi(env_mode + 5) == 512) ? 512 : 256);
    const int TILE_LM = pl.wat ? WTILE_LMKS : TILE_LMKS;
    pl.tile = TILE;
    const size_t shmem = pl.wat
        ? sizeof(double) * ((size_t)((acc_doubles + 1) & ~1) + WAT_WAVES * WAVE_LDS_DOUBLES + 2)
        : sizeof(double) * ((size_t)((acc_doubles + 1) & ~1) + TILE * 9 + TILE_LMKS * 12);
    if (shmem > (size_t)LDS_BYTES || p.F == 0 || p.C == 0 || p.C >= (1 << META_CAM_BITS)) return 0;   // general sweep instead

    std::vector<int4> tiles;
    std::vector<int32_t> big;
    int cur_f0 = 0, cur_l0 = 0, cur_nf = 0, cur_nl = 0;
    auto flush = [&]() {
        if (cur_nl > 0) tiles.push_back(make_int4(cur_f0, cur_l0, cur_nf | (cur_nl << 16), 0));
        cur_nf = 0; cur_nl = 0;
    };
    for (int l = 0; l < p.L; ++l) {
        const int deg = lptr[l + 1] - lptr[l];
        if (deg > TILE) {
            flush();
            for (int o = 0; o < deg; o += TILE)
                tiles.push_back(make_int4(lptr[l] + o, l, std::min(TILE, deg - o), 0));
            big.push_back(l);
            continue;
        }
        if (cur_nl > 0 && (cur_nf + deg > TILE || cur_nl == TILE_LM)) flush();
        if (cur_nl == 0) { cur_f0 = lptr[l]; cur_l0 = l; }
        cur_nf += deg; cur_nl += 1;
    }
    flush();

    std::vector<unsigned> meta((size_t)p.Fp, 0u);
    std::vector<int32_t> stamp((size_t)p.C, -1), count((size_t)p.C, 0);
    for (size_t t = 0; t < tiles.size(); ++t) {
        int4 &td = tiles[t];
        const int nf = td.z & 0xffff;
        int mr = 0;
        for (int i = 0; i < nf; ++i) {
            const int f = td.x + i, c = fcam[f];
            if (stamp[c] != (int32_t)t) { stamp[c] = (int32_t)t; count[c] = 0; }
            state[f] = (int32_t)(((uint32_t)state[f] & ~(STATE_RANK_MASK << 2)) | ((uint32_t)count[c] << 2));   // rank bits of the state word
            // landmark slot inside the tile: rank of its landmark among the tile's landmarks (0 for chunk tiles)
            mr = std::max(mr, count[c]);
            count[c]++;
        }
        td.w = mr;
        const int nl = td.z >> 16;
        for (int j = 0; j < nl; ++j)
            for (int f = lptr[td.y + j]; f < lptr[td.y + j + 1]; ++f) meta[f] = (unsigned)fcam[f] | ((unsigned)j << META_CAM_BITS);
        if (nl == 0)
            for (int i = 0; i < nf; ++i) meta[td.x + i] = (unsigned)fcam[td.x + i];
    }
    pl.n_tiles = (int)tiles.size();
    pl.n_blocks = std::max(1, std::min(pl.n_tiles, n_cus));
    std::vector<int32_t> blk((size_t)pl.n_blocks + 1);
    for (int b = 0; b <= pl.n_blocks; ++b) blk[b] = (int32_t)((int64_t)b * pl.n_tiles / pl.n_blocks);

    int4 *d_tiles = nullptr; int *d_blk = nullptr; unsigned *d_meta = nullptr; double *d_bp = nullptr;
    if (fused_upload(pl, &d_tiles, tiles.data(), tiles.size(), stream)) return -1;
    if (fused_upload(pl, &d_blk, blk.data(), blk.size(), stream)) return -1;
    if (fused_upload(pl, &d_meta, meta.data(), meta.size(), stream)) return -1;
    if (fused_upload<double>(pl, &d_bp, nullptr, (size_t)pl.n_blocks * acc_doubles, stream)) return -1;
    pl.n_big = (int)big.size();
    if (pl.n_big && fused_upload(pl, &pl.d_big, big.data(), big.size(), stream)) return -1;
    const char *env_dbg = getenv("GBP_FUSED_DBG");
    pl.args = FusedArgs{d_meta, d_bp, acc_doubles, env_dbg ? atoi(env_dbg) : 0};
    pl.d_tiles = d_tiles; pl.d_blk = d_blk;
    pl.shmem = shmem;

#define GBP_SET_SHMEM(K)                                                                                              \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize,           \
                            (int)shmem) != hipSuccess) return -1;
#define GBP_SET_ALL(L)                                                                                                \
    GBP_SET_SHMEM((k_sweep_fused<L, true, 256>)) GBP_SET_SHMEM((k_sweep_fused<L, false, 256>))                         \
    GBP_SET_SHMEM((k_sweep_fused<L, false, 512>))
    GBP_SET_ALL(0) GBP_SET_ALL(1) GBP_SET_ALL(2)
#undef GBP_SET_ALL
    GBP_SET_SHMEM((k_sweep_wat<0, WAT_WAVES>)) GBP_SET_SHMEM((k_sweep_wat<1, WAT_WAVES>)) GBP_SET_SHMEM((k_sweep_wat<2, WAT_WAVES>))
#undef GBP_SET_SHMEM
    const char *env = getenv("GBP_FUSED_PREFETCH");
    pl.prefetch = !(env && env[0] == '0');
    pl.enabled = true;
    return 0;
}

// returns 0 or a hipError_t value
inline int fused_launch(FusedPlan &pl, const Params &p0, int robustify, int local_relin, double *partial, hipStream_t stream,
                        hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    Params p = p0;
    p.robustify = robustify; p.local_relin = local_relin;
    const dim3 grid(pl.n_blocks), block(pl.wat ? WAT_WAVES * 64 : pl.tile);
    if (e0) (void)hipEventRecord(e0, stream);
#define GBP_LAUNCH(L)                                                                                                        \
    if (pl.wat) hipLaunchKernelGGL((k_sweep_wat<L, WAT_WAVES>), grid, block, pl.shmem, stream, p, pl.args, pl.d_tiles, pl.d_blk); \
    else if (pl.tile == 512) hipLaunchKernelGGL((k_sweep_fused<L, false, 512>), grid, block, pl.shmem, stream, p, pl.args, pl.d_tiles, pl.d_blk); \
    else if (pl.prefetch) hipLaunchKernelGGL((k_sweep_fused<L, true, 256>), grid, block, pl.shmem, stream, p, pl.args, pl.d_tiles, pl.d_blk); \
    else hipLaunchKernelGGL((k_sweep_fused<L, false, 256>), grid, block, pl.shmem, stream, p, pl.args, pl.d_tiles, pl.d_blk);
    switch (p.loss) {
    case 0: GBP_LAUNCH(0) break;
    case 1: GBP_LAUNCH(1) break;
    default: GBP_LAUNCH(2) break;
    }
#undef GBP_LAUNCH
    if (e1) (void)hipEventRecord(e1, stream);
    if (pl.n_big) hipLaunchKernelGGL(k_lmk_belief_list, dim3((pl.n_big + 63) / 64), dim3(64), 0, stream, p, pl.d_big, pl.n_big);
    hipLaunchKernelGGL(k_cam_reduce_blocks, dim3((pl.args.acc_doubles + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream,
                       pl.args.block_partials, pl.n_blocks, pl.args.acc_doubles, partial);
    return (int)hipGetLastError();
}

}  // namespace gbp
