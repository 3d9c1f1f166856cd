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
//     pre-computed rank (round r: lanes with rank r add, then a barrier) -> no atomics, bitwise
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

constexpr int TILE = 256;
constexpr int LDS_BYTES = 160 * 1024;

struct FusedArgs {
    const int4 *tiles;          // {f0, l0, nf | nl << 16, max rank}
    const int *blk_begin;       // [n_blocks + 1] tile ranges
    const unsigned char *rank;  // [Fp] per factor: index among same-camera factors of its tile
    double *block_partials;     // [n_blocks][C*27]
    int acc_doubles;            // C*27
};

struct Stream {                 // everything a factor streams from HBM each sweep
    double x0[9], z[2], eC[6], MC[21], eL[3], ML[6], avar;
    int st, cam, lmk, rank;
};

template <int LOSS>
GBP_DEV void load_stream(const Params &p, const FusedArgs &a, int f, Stream &s)
{
    const size_t Fp = (size_t)p.Fp;
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
    s.st = p.state[f];
    s.cam = p.fcam[f];
    s.lmk = p.flmk[f];
    s.rank = a.rank[f];
    s.avar = (LOSS != 0) ? p.avar[f] : p.sigma2;
}

template <int LOSS, bool PREFETCH>
__global__ __launch_bounds__(TILE, 1) void k_sweep_fused(Params p, FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *acc = smem;                                   // [C][27] camera accumulators
    double *s_ml = smem + ((a.acc_doubles + 1) & ~1);     // [TILE][9] new landmark messages of the tile
    double *s_lb = s_ml + TILE * 9;                       // [TILE][12] landmark belief (9) + mean (3) of the tile
    const int tid = threadIdx.x;
    const size_t Fp = (size_t)p.Fp, Lp = (size_t)p.Lp;
    for (int i = tid; i < a.acc_doubles; i += TILE) acc[i] = 0.0;

    const int tb = a.blk_begin[blockIdx.x], te = a.blk_begin[blockIdx.x + 1];
    Stream cur, nxt;
    double lbn[12];                                       // next tile's landmark belief/mean (lanes < nl)
    int4 td = make_int4(0, 0, 0, 0), tdn = make_int4(0, 0, 0, 0);
    if (tb < te) {
        td = a.tiles[tb];
        if (tid < (td.z & 0xffff)) load_stream<LOSS>(p, a, td.x + tid, cur);
        if (tid < (td.z >> 16)) {
#pragma unroll
            for (int k = 0; k < 9; ++k) lbn[k] = p.lbel[k * Lp + td.y + tid];
#pragma unroll
            for (int k = 0; k < 3; ++k) lbn[9 + k] = p.lmu[k * Lp + td.y + tid];
        }
    }
    __syncthreads();

    for (int t = tb; t < te; ++t) {
        const int f0 = td.x, l0 = td.y, nf = td.z & 0xffff, nl = td.z >> 16, maxrank = td.w;
        const bool active = tid < nf;
        // stage this tile's landmark beliefs for its factor lanes
        if (tid < nl) {
#pragma unroll
            for (int k = 0; k < 12; ++k) s_lb[tid * 12 + k] = lbn[k];
        }
        if (PREFETCH && t + 1 < te) {
            tdn = a.tiles[t + 1];
            if (tid < (tdn.z & 0xffff)) load_stream<LOSS>(p, a, tdn.x + tid, nxt);
            if (tid < (tdn.z >> 16)) {
#pragma unroll
                for (int k = 0; k < 9; ++k) lbn[k] = p.lbel[k * Lp + tdn.y + tid];
#pragma unroll
                for (int k = 0; k < 3; ++k) lbn[9 + k] = p.lmu[k * Lp + tdn.y + tid];
            }
        }
        __syncthreads();

        if (active) {
            const int f = f0 + tid;
            double etaC[6], lamC[21], muC[6], etaL[3], lamL[6], muL[3];
            load_cam_record(p.cbel + (size_t)cur.cam * CAMREC, etaC, lamC, muC);
            if (nl > 0) {
                const double *lb = s_lb + (cur.lmk - l0) * 12;
#pragma unroll
                for (int k = 0; k < 3; ++k) etaL[k] = lb[k];
#pragma unroll
                for (int k = 0; k < 6; ++k) lamL[k] = lb[3 + k];
#pragma unroll
                for (int k = 0; k < 3; ++k) muL[k] = lb[9 + k];
            } else {                                      // chunk of an over-sized landmark
#pragma unroll
                for (int k = 0; k < 3; ++k) etaL[k] = p.lbel[k * Lp + cur.lmk];
#pragma unroll
                for (int k = 0; k < 6; ++k) lamL[k] = p.lbel[(3 + k) * Lp + cur.lmk];
#pragma unroll
                for (int k = 0; k < 3; ++k) muL[k] = p.lmu[k * Lp + cur.lmk];
            }
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
        __syncthreads();

        // landmark beliefs of the tile: prior + messages in adj_factors order (gbp.py:182-193)
        if (tid < nl) {
            const int l = l0 + tid;
            const int r0 = p.lptr[l] - f0, r1 = p.lptr[l + 1] - f0;
            double b[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) b[k] = p.lprior[k * Lp + l];
            for (int r = r0; r < r1; ++r) {
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

        // camera accumulation, same-camera lanes serialised by rank
        for (int r = 0; r <= maxrank; ++r) {
            if (active && cur.rank == r) {
                double *dst = acc + cur.cam * 27;
#pragma unroll
                for (int k = 0; k < 6; ++k) dst[k] += cur.eC[k];
#pragma unroll
                for (int k = 0; k < 21; ++k) dst[6 + k] += cur.MC[k];
            }
            __syncthreads();
        }

        if (t + 1 < te) {
            if (PREFETCH) {
                cur = nxt;
                td = tdn;
            } else {
                td = a.tiles[t + 1];
                if (tid < (td.z & 0xffff)) load_stream<LOSS>(p, a, td.x + tid, cur);
                if (tid < (td.z >> 16)) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) lbn[k] = p.lbel[k * Lp + td.y + tid];
#pragma unroll
                    for (int k = 0; k < 3; ++k) lbn[9 + k] = p.lmu[k * Lp + td.y + tid];
                }
            }
        }
    }
    double *out = a.block_partials + (size_t)blockIdx.x * a.acc_doubles;
    for (int i = tid; i < a.acc_doubles; i += TILE) out[i] = acc[i];
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
    int n_tiles = 0, n_blocks = 0, n_big = 0;
    size_t shmem = 0;
    FusedArgs args{};
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
                      hipStream_t stream, int n_cus)
{
    const int acc_doubles = p.C * 27;
    const size_t shmem = sizeof(double) * ((size_t)((acc_doubles + 1) & ~1) + TILE * 9 + TILE * 12);
    if (shmem > (size_t)LDS_BYTES || p.F == 0 || p.C == 0) return 0;          // general sweep instead

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
        if (cur_nl > 0 && (cur_nf + deg > TILE || cur_nl == TILE)) flush();
        if (cur_nl == 0) { cur_f0 = lptr[l]; cur_l0 = l; }
        cur_nf += deg; cur_nl += 1;
    }
    flush();

    std::vector<unsigned char> rank((size_t)p.Fp, 0);
    std::vector<int32_t> stamp((size_t)p.C, -1), count((size_t)p.C, 0);
    for (size_t t = 0; t < tiles.size(); ++t) {
        int4 &td = tiles[t];
        const int nf = td.z & 0xffff;
        int mr = 0;
        for (int i = 0; i < nf; ++i) {
            const int f = td.x + i, c = fcam[f];
            if (stamp[c] != (int32_t)t) { stamp[c] = (int32_t)t; count[c] = 0; }
            rank[f] = (unsigned char)count[c];
            mr = std::max(mr, count[c]);
            count[c]++;
        }
        td.w = mr;
    }
    pl.n_tiles = (int)tiles.size();
    pl.n_blocks = std::max(1, std::min(pl.n_tiles, n_cus));
    std::vector<int32_t> blk((size_t)pl.n_blocks + 1);
    for (int b = 0; b <= pl.n_blocks; ++b) blk[b] = (int32_t)((int64_t)b * pl.n_tiles / pl.n_blocks);

    int4 *d_tiles = nullptr; int *d_blk = nullptr; unsigned char *d_rank = nullptr; double *d_bp = nullptr;
    if (fused_upload(pl, &d_tiles, tiles.data(), tiles.size(), stream)) return -1;
    if (fused_upload(pl, &d_blk, blk.data(), blk.size(), stream)) return -1;
    if (fused_upload(pl, &d_rank, rank.data(), rank.size(), stream)) return -1;
    if (fused_upload<double>(pl, &d_bp, nullptr, (size_t)pl.n_blocks * acc_doubles, stream)) return -1;
    pl.n_big = (int)big.size();
    if (pl.n_big && fused_upload(pl, &pl.d_big, big.data(), big.size(), stream)) return -1;
    pl.args = FusedArgs{d_tiles, d_blk, d_rank, d_bp, acc_doubles};
    pl.shmem = shmem;

#define GBP_SET_SHMEM(K)                                                                                              \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize,           \
                            (int)shmem) != hipSuccess) return -1;
    GBP_SET_SHMEM((k_sweep_fused<0, true>)) GBP_SET_SHMEM((k_sweep_fused<1, true>)) GBP_SET_SHMEM((k_sweep_fused<2, true>))
    GBP_SET_SHMEM((k_sweep_fused<0, false>)) GBP_SET_SHMEM((k_sweep_fused<1, false>)) GBP_SET_SHMEM((k_sweep_fused<2, false>))
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
    const dim3 grid(pl.n_blocks), block(TILE);
    if (e0) (void)hipEventRecord(e0, stream);
#define GBP_LAUNCH(L)                                                                                               \
    if (pl.prefetch) hipLaunchKernelGGL((k_sweep_fused<L, true>), grid, block, pl.shmem, stream, p, pl.args);       \
    else hipLaunchKernelGGL((k_sweep_fused<L, false>), grid, block, pl.shmem, stream, p, pl.args);
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
