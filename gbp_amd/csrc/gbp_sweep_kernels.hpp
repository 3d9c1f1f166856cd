// gbp_sweep_kernels.hpp -- the kernels of the general sweep, the stage-wise calls, the camera finish / peer exchange and the diagnostics
// (launched by gbp_capi_sweep.hip only; layout and device helpers: gbp_kernels.hpp; the fused sweep: gbp_fused.hpp)
#pragma once
#include "gbp_kernels.hpp"

namespace gbp {

// ------------------------------------------------------------- general sweep, tile version --
// One WAVE per tile: the per-factor part of synchronous_iteration (both messages computed from the OLD messages and
// committed together, Factor.compute_messages gbp.py:334-373), plus what the tile structure gives for free when the
// camera table of the fused sweep does not fit the LDS (C > 516):
//   * the landmarks a tile owns get their beliefs from the same wave (new messages through LDS, prior + sum in
//     adj_factors order, 3x3 solve) -- no second pass over the messages (k_lmk_belief re-reads and re-linearises them);
//   * what rebuilds the message to the camera (x0 9 | q_C 2 | W 3: eta = Jc^T q_C, Lambda = Jc^T W Jc with Jc at x0 -- 14
//     doubles instead of the dense 27) is written to cstage[cpos[slot]], i.e. in the camera's own
//     adj_factors order, so k_cam_partial_staged reads one contiguous run per camera instead of gathering 16-byte
//     pieces and rebuilding Jacobians (k_cam_partial fetches 562 MB per sweep at 1M factors; this is 216 + 216 MB).
template <int LOSS, bool XTRA>
__global__ __launch_bounds__(BLOCK, 1) void k_factor_tile(Params p)
{
    __shared__ __attribute__((aligned(16))) double wls[BLOCK / 64][WTILE * CSTAGE_ROW];  // per wave: [64][9] landmark messages, then [64][20] camera-message rows
    __shared__ int wps[BLOCK / 64][WTILE];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int t = blockIdx.x * (BLOCK / 64) + wave;
    if (t >= p.T) return;                                   // whole wave
    if (p.reverse_walk) t = p.T - 1 - t;
    double *wl = wls[wave];
    int *wp = wps[wave];
    const int4 td = p.tiles[t];
    const int l0 = td.x, nl = td.y, nf = td.z;
    const bool active = lane < nf;
    const int slot = t * WTILE + lane;
    double srow[XTRA ? CSTAGE_ROW : CSTAGE_PLAIN];          // x0 | q_C | W (| remainder, or two pad doubles) of this lane's factor AFTER the sweep
    if (active) {
        const unsigned meta = slot_meta(p, slot);
        const int cam = (int)(meta >> META_LMK_BITS), lmk = l0 + (int)(meta & ((1u << META_LMK_BITS) - 1u));
        double x0[9], z[2], qC[2], qL[2], WC[3], VL[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) x0[k] = p.lin[lin_at(slot, ROW_X0 + k)];
        z[0] = p.lin[lin_at(slot, ROW_Z)]; z[1] = p.lin[lin_at(slot, ROW_Z + 1)];
#pragma unroll
        for (int k = 0; k < 2; ++k) { qC[k] = p.msg[msg_at(slot, ROW_QC + k)]; qL[k] = p.msg[msg_at(slot, ROW_QL + k)]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) WC[k] = p.msg[msg_at(slot, ROW_WC + k)];
#pragma unroll
        for (int k = 0; k < 3; ++k) VL[k] = p.msg[msg_at(slot, ROW_VL + k)];
        int st = slot_state(p, slot);
        const int st_in = st;
        double avar = (LOSS != 0) ? p.avar[slot] : p.sigma2;
        double muC[6], PC[21], muL[3];
        load_cam_record(p.cbel + (size_t)cam * CAMREC, muC, PC);
        const double *lr = p.lrec + (size_t)lmk * LREC;
#pragma unroll
        for (int k = 0; k < 3; ++k) muL[k] = lr[LR_MU + k];

        double eCn[6], eLn[3], MCn[21], MLn[6];
        const bool relin = factor_core<LOSS, XTRA>(p, x0, z, st, avar, muC, PC, muL,
                                                   [lr](double (&c)[6]) {
#pragma unroll
                                                       for (int k = 0; k < 6; ++k) c[k] = lr[LR_COV + k];
                                                   },
                                                   [&p, slot](const double (&x)[9]) {
#pragma unroll
                                                       for (int k = 0; k < 9; ++k) p.lin[lin_at(slot, ROW_X0 + k)] = x[k];
                                                   },
                                                   qC, qL, WC, VL, eCn, eLn, MCn, MLn,
                                                   XTRA ? p.xtra + (size_t)slot * XTRA_ROW : nullptr);
        {
            const unsigned long long rb = __ballot(relin);
            if (rb != 0ull && lane == __ffsll((long long)rb) - 1) relin_add(p, __popcll(rb));
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) { p.msg[msg_at(slot, ROW_QC + k)] = qC[k]; p.msg[msg_at(slot, ROW_QL + k)] = qL[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) p.msg[msg_at(slot, ROW_WC + k)] = WC[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) p.msg[msg_at(slot, ROW_VL + k)] = VL[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) wl[lane * 9 + k] = eLn[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) wl[lane * 9 + 3 + k] = MLn[k];
        if (st != st_in) set_slot_state(p, slot, st);
        if (LOSS != 0) p.avar[slot] = avar;
        wp[lane] = p.cpos[slot];
#pragma unroll
        for (int k = 0; k < 9; ++k) srow[k] = x0[k];
        srow[9] = qC[0]; srow[10] = qC[1];
#pragma unroll
        for (int k = 0; k < 3; ++k) srow[11 + k] = WC[k];
        if (XTRA) {
#pragma unroll
            for (int k = 0; k < 6; ++k) srow[CSTAGE_USED + k] = p.xtra[(size_t)slot * XTRA_ROW + k];
        }
    }
    if (p.stage & STAGE_NO_BELIEFS) return;                 // compute_all_messages on its own (gbp.py:46-54): whole wave
    // VariableNode.update_belief gbp.py:176-198 for the tile's landmarks: nine lanes per landmark
    LmkPre pre;
    lmk_prefetch(p, lane, t, l0, nl, pre);
    wave_lds_sync();                                        // the wave's LDS writes are done (one wave: no barrier needed)
    tile_landmark_beliefs(p, wl, lane, t, l0, nl, pre);
    // camera-message rows -> cstage through LDS: a lane-per-factor store would touch 64 different lines per instruction;
    // transposed, whole rows go out, 16 bytes per lane
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // landmark phase has read the [64][9] messages
    constexpr int R = XTRA ? CSTAGE_ROW : CSTAGE_PLAIN;
    if (!XTRA) { srow[CSTAGE_USED] = 0.0; srow[CSTAGE_USED + 1] = 0.0; }
    if (active) {
#pragma unroll
        for (int k = 0; k < R; ++k) wl[lane * R + k] = srow[k];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        constexpr int R2 = R / 2, PER2 = 64 / R2;           // 16 bytes per lane: eight whole 128-byte lines per instruction (six 160-byte rows with xtra)
        const int g = lane / R2, k = 2 * (lane - g * R2);
        if (g < PER2) {
            for (int f = g; f < nf; f += PER2)
                *reinterpret_cast<double2 *>(p.cstage + (size_t)wp[f] * R + k) = *reinterpret_cast<const double2 *>(wl + f * R + k);
        }
    }
}

// One workgroup per camera: partial[c][27] = sum of the messages of its factors, rebuilt from the staged rows (contiguous per
// camera, in the reference's adj_factors order): every thread linearises its factors (every 256th of the run), adds their
// eta = Jc^T q_C (+ remainder) and Lambda = Jc^T W Jc into 27 registers, then the block adds the threads up in a fixed order
// (shuffle tree, then the four waves) -- bitwise reproducible.
// finish != 0 (single GPU: nothing to exchange): the camera belief is completed here -- prior + sum, 6x6 solve (gbp.py:182-193) --
// instead of in a dependent k_cam_finish launch.
// the sum of camera c's messages from its staged rows: entry tid (< 27) in threads 0..26 (undefined elsewhere); red = [NT / 64][27] of LDS.
// The caller synchronises the block before red is used again.
template <int NT>
GBP_DEV double cam_staged_sum(const Params &p, int c, double (*red)[27])
{
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    const int e1 = p.cptr[c + 1];
    for (int e = p.cptr[c] + threadIdx.x; e < e1; e += NT) {
        const double2 *row = reinterpret_cast<const double2 *>(p.cstage + (size_t)e * p.crow);
        double v[CSTAGE_ROW];
#pragma unroll
        for (int i = 0; i < CSTAGE_USED / 2; ++i) { const double2 t = row[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
        double x0[9], Jc[2][6], Jl[2][3], h[2], MC[21];
#pragma unroll
        for (int k = 0; k < 9; ++k) x0[k] = v[k];
        linearise(x0, p.K, Jc, Jl, h);
        const double W[3] = {v[11], v[12], v[13]};
#pragma unroll
        for (int k = 0; k < 21; ++k) MC[k] = 0.0;
        rank2_update<6>(MC, Jc[0], Jc[1], W, 1.0);
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += Jc[0][k] * v[9] + Jc[1][k] * v[10];
        if (p.xtra) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { const double2 t = row[CSTAGE_USED / 2 + i]; acc[2 * i] += t.x; acc[2 * i + 1] += t.y; }
        }
#pragma unroll
        for (int k = 0; k < 21; ++k) acc[6 + k] += MC[k];
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        acc[k] = v;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 27; ++k) red[wave][k] = acc[k];
    }
    __syncthreads();
    double s2 = 0.0;
    if (threadIdx.x < 27) {
        s2 = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) s2 += red[w][threadIdx.x];
    }
    return s2;
}

template <int NT>
__global__ __launch_bounds__(NT) void k_cam_partial_staged(Params p, double *__restrict__ partial, int finish)
{
    __shared__ double red[NT / 64][27];
    __shared__ double tot[27];
    const int c = p.reverse_walk ? p.C - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const double pri = threadIdx.x < 27 ? p.cprior[(size_t)c * 27 + threadIdx.x] : 0.0;      // (asked for ahead of the rows: one round trip less in the tail)
    const double s2 = cam_staged_sum<NT>(p, c, red);
    if (threadIdx.x < 27) {
        partial[(size_t)c * 27 + threadIdx.x] = s2;
        tot[threadIdx.x] = s2 + pri;
    }
    if (!finish) return;
    __syncthreads();
    double *rec = p.cbel + (size_t)c * CAMREC;
    if (threadIdx.x >= NT - 27) p.cbelief[(size_t)c * CBEL + threadIdx.x - (NT - 27)] = tot[threadIdx.x - (NT - 27)];
    if (threadIdx.x < 7) {
        double v[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = tot[k];
        cam_belief_store(v, rec, threadIdx.x);
    }
}

// The general sweep under the peer-store exchange, everything after the factor kernel in ONE launch (the staged counterpart of
// k_cam_reduce_xchg, gbp_fused.hpp): a grid of persistent workgroups, never larger than what is resident at once, first sums and
// PUSHES all of its cameras (staged rows -> 27 sums -> row c of every rank's mailbox), then finishes them, one wave per camera
// (take the n_ranks rows c as they arrive, add the parts in rank order, prior, mean | covariance).  Same sums, bitwise, as
// k_cam_partial_staged + k_peer_push + k_cam_finish, which remain the path of logical ranks that share one device (rendezvous hook).
template <int NT>
__global__ __launch_bounds__(NT) void k_cam_staged_xchg(Params p, double *__restrict__ partial, PeerOut peer, PeerWait wait)
{
    __shared__ double red[NT / 64][27];
    const int tid = threadIdx.x;
    if (wait.clk && blockIdx.x == 0 && tid == 0) *wait.clk = (unsigned long long)wall_clock64();
    for (int b = blockIdx.x; b < p.C; b += gridDim.x) {
        const int c = p.reverse_walk ? p.C - 1 - b : b;
        const double s = cam_staged_sum<NT>(p, c, red);
        if (tid < 64) {
            if (tid < 27) partial[(size_t)c * 27 + tid] = s;
            peer_push_row(peer, c, s, tid);
        }
        __syncthreads();                                     // red is overwritten by the next camera
    }
    const int wave = tid >> 6, lane = tid & 63;
    for (int b = blockIdx.x + wave * gridDim.x; b < p.C; b += (NT / 64) * gridDim.x)
        cam_finish_wave(p, nullptr, peer.n, 0, wait, p.reverse_walk ? p.C - 1 - b : b, lane);
}

__global__ __launch_bounds__(BLOCK) void k_lmk_belief(Params p)
{
    const int l = blockIdx.x * BLOCK + threadIdx.x;
    if (l < p.L) landmark_belief_from_hbm(p, l);
}

// beliefs of the landmarks that span more than one tile, after a sweep (gbp_kernels.hpp: finish_landmark_parts); one thread per tile
__global__ __launch_bounds__(BLOCK) void k_lmk_finish_parts(Params p)
{
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t < p.T) finish_landmark_parts(p, t);
}

// One workgroup per camera: sum of the messages of its factors (gathered through cadj), WITHOUT the
// prior, into partial[c][27].  Fixed-shape reduction -> bitwise reproducible.
__global__ __launch_bounds__(BLOCK) void k_cam_partial(Params p, double *__restrict__ partial)
{
    __shared__ double red[BLOCK / 64][27];
    const int c = blockIdx.x;
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    const int e1 = p.cptr[c + 1];
    for (int e = p.cptr[c] + threadIdx.x; e < e1; e += BLOCK) {
        const int s = p.cadj[e];
        double eC[6], MC[21], eL[3], ML[6];
        dense_messages(p, s, eC, MC, eL, ML);
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += eC[k];
#pragma unroll
        for (int k = 0; k < 21; ++k) acc[6 + k] += MC[k];
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        acc[k] = v;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 27; ++k) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        double s = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < BLOCK / 64; ++w) s += red[w][threadIdx.x];
        partial[(size_t)c * 27 + threadIdx.x] = s;
    }
}


// Self-test of the peer-store exchange, run by every rank after gbp_ba_peer_connect and before the first sweep (gbp_ba_peer_selftest): ONE
// wave stores a probe row -- 27 values that depend on (sender, entry, test number) -- into the probe area of EVERY rank's mailbox
// exactly the way the sweep's rows travel (write-through stores into slots that hold PEER_EMPTY), then polls the probe rows of all
// ranks in its own mailbox until none of their slots is empty, compares every entry and empties them again.  out[0] |= 1: a rank's row
// did not arrive in time, |= 2: it arrived with wrong contents; out[1] = the (lowest) rank concerned.  A pair of devices whose mapping or
// atomics do not work shows up here, with a name, instead of as a time-out or a wrong belief in the middle of a run.
GBP_HD double peer_probe_value(int src, int k, unsigned long long seq) { return (double)(((long long)(src + 1) << 20) + ((long long)k << 12) + (long long)(seq & 0xfffu)); }     // (an integer: exact however it is evaluated)
__global__ __launch_bounds__(64) void k_peer_selftest(PeerOut peer, double *mine, int rank, long long timeout_ticks, int *out)
{
    const int lane = threadIdx.x;
    for (int r = 0; r < peer.n; ++r)
        if (lane < 27) peer_store(peer.dst[r] + lane, peer_probe_value(rank, lane, peer.seq));
    unsigned long long late = 0ull, wrong = 0ull;
    const long long t0 = wall_clock64();
    for (int r = 0; r < peer.n; ++r) {
        double v = 0.0;
        bool arrived = true;
        for (;;) {
            bool missing = false;
            if (lane < 27) { v = peer_load(mine + (size_t)r * PEER_ROW + lane); missing = peer_is_empty(v); }
            if (!__any(missing)) break;
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > timeout_ticks) { arrived = false; break; }
        }
        if (!__all(arrived)) { late |= 1ull << r; continue; }
        if (__ballot(lane < 27 && v != peer_probe_value(r, lane, peer.seq))) wrong |= 1ull << r;
        if (lane < 27) peer_store(mine + (size_t)r * PEER_ROW + lane, peer_empty_value());
    }
    if (lane == 0 && (late | wrong)) {
        out[0] = (late ? 1 : 0) | (wrong ? 2 : 0);
        out[1] = __ffsll((long long)(late | wrong)) - 1;
    }
}

// general path / update_all_beliefs: the partial sums already sit in `partial` (C*27): one wave per camera moves its row
__global__ __launch_bounds__(BLOCK) void k_peer_push(const double *__restrict__ partial, int C, PeerOut peer)
{
    const int lane = threadIdx.x & 63, c = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (c >= C) return;                                     // whole wave
    peer_push_row(peer, c, lane < 27 ? partial[(size_t)c * 27 + lane] : 0.0, lane);
}


constexpr int FINISH_BLOCK = 256;
__global__ __launch_bounds__(FINISH_BLOCK) void k_cam_finish(Params p, const double *gathered, int n_parts, size_t part_stride, PeerWait wait)
{
    if (wait.clk && blockIdx.x == 0 && threadIdx.x == 0) *wait.clk = (unsigned long long)wall_clock64();
    const int lane = threadIdx.x & 63, c = blockIdx.x * (FINISH_BLOCK / 64) + (threadIdx.x >> 6);
    if (c >= p.C) return;                                   // whole wave
    cam_finish_wave(p, gathered, n_parts, part_stride, wait, c, lane);
}

// instrumented runs: one stamp of the device's constant-rate clock (the calibration of gbp_ba_set_kernel_timing)
__global__ void k_clk_stamp(unsigned long long *out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = (unsigned long long)wall_clock64();
}

// instrumented runs: the stamp ring starts empty
__global__ void k_clk_init(unsigned long long *clk, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) clk[i] = 0ull;
}

// ----------------------------------------------------------------------------- diagnostics --
// Factor.compute_residual at the current belief means (gbp.py:251-259); per-workgroup partial sums of
// ||r|| (BAFactorGraph.are gbp_ba.py:61-69) and 0.5||r||^2/adaptive_var (FactorGraph.energy gbp.py:36-44).
__global__ __launch_bounds__(BLOCK) void k_residual(Params p, double *__restrict__ partials)
{
    __shared__ double red[BLOCK / 64][2];
    const int slot = blockIdx.x * BLOCK + threadIdx.x;
    double nr = 0.0, en = 0.0;
    int cam, lmk;
    if (slot < p.T * WTILE && slot_info(p, slot, cam, lmk)) {
        double x[9], h[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] = p.cbel[(size_t)cam * CAMREC + CAM_MU + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[6 + k] = p.lrec[(size_t)lmk * LREC + LR_MU + k];
        project(x, p.K, h);
        const double r0 = h[0] - p.lin[lin_at(slot, ROW_Z)], r1 = h[1] - p.lin[lin_at(slot, ROW_Z + 1)];
        nr = sqrt(r0 * r0 + r1 * r1);
        const double av = slot_avar(p, slot);
        en = 0.5 * (nr * nr) / av;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { nr += __shfl_down(nr, off, 64); en += __shfl_down(en, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = nr; red[threadIdx.x >> 6][1] = en; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) { a += red[w][0]; b += red[w][1]; }
        partials[2 * (size_t)blockIdx.x] = a;
        partials[2 * (size_t)blockIdx.x + 1] = b;
    }
}


// ---------------------------------------------------------------- stage-wise entry points --
// The reference lets a caller run the stages of a sweep one by one (gbp.py:46-84).  robustify and the relinearisation DECISION
// are state-word / variance updates; the relinearisation itself is deferred to the next message computation (state word header).

// FactorGraph.robustify_all_factors (gbp.py:82-84 -> Factor.robustify_loss gbp.py:296-332): the adaptive variance and the robust
// flag from the residual at the linearisation point; eta_f / Lambda_f are rebuilt from (x0, z, variance) wherever they are needed.
template <int LOSS>
__global__ __launch_bounds__(BLOCK) void k_stage_robustify(Params p)
{
    const int slot = blockIdx.x * BLOCK + threadIdx.x;
    int cam, lmk;
    if (slot >= p.T * WTILE || !slot_info(p, slot, cam, lmk)) return;
    double x0[9], h0[2];
    effective_linpoint(p, slot, x0);
    project(x0, p.K, h0);
    int st = slot_state(p, slot);
    bool robust = (st & 2) != 0;
    const double avar = robust_variance(LOSS, p.sigma2, p.nstds, p.lin[lin_at(slot, ROW_Z)] - h0[0], p.lin[lin_at(slot, ROW_Z + 1)] - h0[1], robust);
    p.avar[slot] = avar;
    set_slot_state(p, slot, (st & ~2) | (robust ? 2 : 0));
}

// FactorGraph.relinearise_factors (gbp.py:64-80), or with mark_all FactorGraph.compute_all_factors (gbp.py:60-62: every factor,
// counters and damping untouched): the decision and its bookkeeping; the move itself is deferred (STATE_PENDING).
__global__ __launch_bounds__(BLOCK) void k_stage_relinearise(Params p, int mark_all)
{
    const int slot = blockIdx.x * BLOCK + threadIdx.x;
    int cam, lmk;
    if (slot >= p.T * WTILE || !slot_info(p, slot, cam, lmk)) return;
    int st = slot_state(p, slot);
    if (mark_all) { set_slot_state(p, slot, st | STATE_PENDING); return; }
    int iters = state_age(st, p.clk - 1);                    // (the host has advanced the clock for this call)
    bool damped = (st & 1) != 0, pending = (st & STATE_PENDING) != 0;
    double d2 = 0.0;
    if (!pending) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double d = p.lin[lin_at(slot, ROW_X0 + k)] - p.cbel[(size_t)cam * CAMREC + CAM_MU + k]; d2 += d * d; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double d = p.lin[lin_at(slot, ROW_X0 + 6 + k)] - p.lrec[(size_t)lmk * LREC + LR_MU + k]; d2 += d * d; }
    }
    if (!pending && sqrt(d2) > p.beta && iters >= p.min_linear) { iters = 0; damped = false; pending = true; }
    else iters = min(iters + 1, ITERS_MAX);
    const int st_new = state_pack(iters, p.clk, state_rank(st), (st & 2) != 0, damped, pending);
    if (st_new != st) set_slot_state(p, slot, st_new);
}

// How many factors would be DAMPED in the very message computation that moves their linearisation point -- a pending relinearisation
// (stage-wise relinearise_factors / compute_all_factors) met by a non-zero eta damping -- if the messages were computed now with
// these flags?  Such a message has a part outside the span of the new Jacobian, which only the dense remainder (Params::xtra) can
// carry; the host allocates it when this count is non-zero (gbp_capi.hip: enable_remainder).  Mirrors factor_decide.
__global__ __launch_bounds__(BLOCK) void k_count_pending_damped(Params p, int local_relin, int no_test, int *__restrict__ out)
{
    const int slot = blockIdx.x * BLOCK + threadIdx.x;
    int cam, lmk;
    bool hit = false;
    if (slot < p.T * WTILE && slot_info(p, slot, cam, lmk)) {
        const int st = slot_state(p, slot);
        if (st & STATE_PENDING) {
            int iters = state_age(st, p.clk);
            if (local_relin && !no_test) iters = min(iters + 1, ITERS_MAX);      // (a pending factor is not tested again: distance 0)
            const bool damped = (st & 1) != 0 || (local_relin && iters == p.num_undamped);
            hit = local_relin ? damped : true;
        }
    }
    const unsigned long long b = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, __popcll(b));
}

// entries of the dense remainder that are not exactly zero (when none is left the handle returns to the fused sweep)
__global__ __launch_bounds__(BLOCK) void k_count_nonzero(const double *__restrict__ x, size_t n, int *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const unsigned long long b = __ballot(i < n && x[i] != 0.0);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, __popcll(b));
}


}  // namespace gbp
