// gbp_capi_sweep.hip -- libgbp_hip.so, every launch of a sweep: the fused sweep and its plan (gbp_fused.hpp), the general sweep and the
// stage-wise calls (gbp_sweep_kernels.hpp), the dense message remainder on demand, the camera finish, diagnostics (ARE / energy) and the
// kernel-time instrumentation.  The kernels are instantiated HERE and nowhere else (gbp_handle.hpp).
#include "gbp_handle.hpp"
#include "gbp_sweep_kernels.hpp"
#include "gbp_fused.hpp"

#include <chrono>
#include <cmath>
#include <thread>

int gbp::plan_fused_sweep(gbp_ba *h, int n_cus) { return fused_plan(h->fused, h->p, h->stream, n_cus, h->wg_win.data(), h->wg_cams.data(), (int)h->wg_win.size()); }
int gbp::fused_max_cams_of_this_build() { return fused_max_cams(); }

// ------------------------------------------------------------------------------ launches --

static bool timing_sample(gbp_ba *h)
{
    if (!h->timing) return false;
    const bool now = (h->timing_tick % h->timing_every) == 0;
    h->timing_tick++;
    return now;
}

static int time_begin(gbp_ba *h)
{
    h->timing_now = timing_sample(h);
    if (!h->timing_now) return GBP_OK;
    if (h->ev_used + 2 > h->ev.size()) {
        for (int i = 0; i < 2; ++i) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->ev.push_back(e); }
    }
    HIPCHK(hipEventRecord(h->ev[h->ev_used], h->stream));
    return GBP_OK;
}

static int time_end(gbp_ba *h)
{
    if (!h->timing_now) return GBP_OK;
    HIPCHK(hipEventRecord(h->ev[h->ev_used + 1], h->stream));
    h->ev_used += 2;
    return GBP_OK;
}

static int launch_factor_stage(gbp_ba *h, int robustify, int local_relin)
{
    Params p = h->p;
    p.robustify = robustify; p.local_relin = local_relin;
    if (!p.T) return GBP_OK;
    const int nb = (p.T + BLOCK / 64 - 1) / (BLOCK / 64);
    h->cstage_x0_ok = false;                                // (this kernel's rows may be the wide ones: the next staged sweep writes whole rows)
    CHK(time_begin(h));
    if (p.xtra) {
        switch (p.loss) {
        case GBP_LOSS_NONE: hipLaunchKernelGGL((k_factor_tile<0, true>), dim3(nb), dim3(BLOCK), 0, h->stream, p); break;
        case GBP_LOSS_HUBER: hipLaunchKernelGGL((k_factor_tile<1, true>), dim3(nb), dim3(BLOCK), 0, h->stream, p); break;
        default: hipLaunchKernelGGL((k_factor_tile<2, true>), dim3(nb), dim3(BLOCK), 0, h->stream, p); break;
        }
    } else {
        switch (p.loss) {
        case GBP_LOSS_NONE: hipLaunchKernelGGL((k_factor_tile<0, false>), dim3(nb), dim3(BLOCK), 0, h->stream, p); break;
        case GBP_LOSS_HUBER: hipLaunchKernelGGL((k_factor_tile<1, false>), dim3(nb), dim3(BLOCK), 0, h->stream, p); break;
        default: hipLaunchKernelGGL((k_factor_tile<2, false>), dim3(nb), dim3(BLOCK), 0, h->stream, p); break;
        }
    }
    CHK(time_end(h));
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

static int launch_lmk_beliefs(gbp_ba *h)
{
    if (!h->p.L) return GBP_OK;
    hipLaunchKernelGGL(k_lmk_belief, dim3(grid_for(h->p.L)), dim3(BLOCK), 0, h->stream, h->p);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

static int launch_cam_partial(gbp_ba *h, double *partial)
{
    if (!h->p.C) return GBP_OK;
    hipLaunchKernelGGL(k_cam_partial, dim3(h->p.C), dim3(BLOCK), 0, h->stream, h->p, partial);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

int gbp::launch_cam_finish(gbp_ba *h, const double *gathered, int n_parts, size_t stride, const PeerWait *wait)
{
    if (!h->p.C) return GBP_OK;
    PeerWait w{};
    if (wait) w = *wait;
    w.clk = h->clk_cur ? h->clk_cur + 4 : nullptr;
    hipLaunchKernelGGL(k_cam_finish, dim3((h->p.C + FINISH_BLOCK / 64 - 1) / (FINISH_BLOCK / 64)), dim3(FINISH_BLOCK), 0, h->stream, h->p, gathered,
                       n_parts, stride, w);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

// camera-major staging of the general sweep, allocated on first use (F x 27 doubles; the slot -> row map cpos is made by the build)
static int ensure_staging(gbp_ba *h)
{
    if (!h->p.cstage || h->cstage_cap < h->p.crow) {
        CHK(dev_alloc(h, &h->p.cstage, std::max<size_t>((size_t)h->p.F, 1) * h->p.crow));
        h->cstage_cap = h->p.crow;
        h->cstage_x0_ok = false;
    }
    return GBP_OK;
}

int gbp::launch_finish_parts(gbp_ba *h, hipStream_t stream)
{
    if (!h->p.parts || !h->p.T) return GBP_OK;
    hipLaunchKernelGGL(k_lmk_finish_parts, dim3(grid_for((size_t)h->p.T)), dim3(BLOCK), 0, stream, h->p);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

// general sweep / update_all_beliefs under the peer-store exchange: the finished partial sums go into every rank's mailbox
static int launch_peer_push(gbp_ba *h, const double *partial, const PeerOut &peer)
{
    if (!h->p.C) return GBP_OK;
    hipLaunchKernelGGL(k_peer_push, dim3((h->p.C + BLOCK / 64 - 1) / (BLOCK / 64)), dim3(BLOCK), 0, h->stream, partial, h->p.C, peer);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

// defer_big: leave the beliefs of the landmarks that span tiles (k_lmk_finish_parts) to the caller, who runs them beside the
// camera exchange (launch_finish_parts)
int gbp::sweep_begin(gbp_ba *h, int with_messages, int robustify, int local_relin, double *partial, int finish, bool *finished, bool defer_big,
                     const PeerOut *peer, const PeerWait *merged)
{
    if (finished) *finished = false;
    h->clk_cur = (h->timing && h->d_clk && h->clk_used < CLK_RING) ? h->d_clk + 6 * (size_t)h->clk_used++ : nullptr;
    if (with_messages) {
        clock_tick(h, local_relin != 0);
        const int slot = (int)(h->sweep_count % RELIN_RING);
        if (slot % (RELIN_RING / 2) == 0)
            HIPCHK(hipMemsetAsync(h->d_relin_ring + (size_t)slot * RELIN_LANES, 0, sizeof(int) * (RELIN_RING / 2) * RELIN_LANES, h->stream));
        h->p.relin_slot = h->d_relin_ring + (size_t)slot * RELIN_LANES;
        h->sweep_count++;
    }
    if (with_messages && h->fused.enabled) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timing_sample(h)) {
            if (h->ev_used + 2 > h->ev.size())
                for (int i = 0; i < 2; ++i) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->ev.push_back(e); }
            e0 = h->ev[h->ev_used]; e1 = h->ev[h->ev_used + 1];
            h->ev_used += 2;
        }
        // Every other sweep walks each workgroup's tile range backwards: what the last sweep touched last is touched first, so
        // whatever part of the state the Infinity Cache still holds is used before it is evicted (GBP_NO_REVERSE: experiment
        // switch).  With arena_reserve this removed the slow mode of the 1M-factor graph (12 of 12 fresh processes at
        // 11.6-12.1k sweeps/s; 7 of 12 at 10.0-10.7k without both).
        static const bool no_rev = getenv("GBP_NO_REVERSE") != nullptr;
        const int reverse = no_rev ? 0 : (int)(h->walk_parity & 1u);
        h->walk_parity ^= 1u;
        h->cstage_x0_ok = false;                            // (a fused sweep moves linearisation points without staging them)
        int rc = fused_launch(h->fused, h->p, robustify, local_relin, partial, h->stream, finish, e0, e1, defer_big, reverse, peer, h->clk_cur, merged);
        if (merged && peer && finished) *finished = true;
        if (rc != 0) return fail(GBP_EHIP, "fused sweep launch failed: %s", hipGetErrorString((hipError_t)rc));
        if (finished && !(merged && peer)) *finished = finish != 0;
        return GBP_OK;
    }
    if (with_messages) {
        // tile sweep: messages + the tiles' landmark beliefs + camera messages staged camera-major.  Every other sweep backwards,
        // like the fused sweep (what the memory-side cache still holds is used first; the results do not depend on the order)
        static const bool no_rev_g = getenv("GBP_NO_REVERSE") != nullptr;
        h->p.reverse_walk = no_rev_g ? 0 : (int)(h->gen_parity & 1u);
        h->gen_parity ^= 1u;
        CHK(ensure_staging(h));
        if (h->p.xtra || getenv("GBP_TILE_KERNEL")) {       // the dense remainder rides in k_factor_tile (one wave per tile)
            h->dominant = "k_factor_tile";
            h->cstage_x0_ok = false;                        // (its rows may be the wide ones: the next staged sweep writes whole rows)
            CHK(launch_factor_stage(h, robustify, local_relin));
        } else {                                             // the persistent loop, staging instead of a camera table
            h->dominant = "k_sweep_staged";
            CHK(time_begin(h));
            const int rc = staged_launch(h->p, robustify, local_relin, h->n_cus, h->p.reverse_walk, h->stream, nullptr, h->cstage_x0_ok ? 0 : 1, &h->staged_attr_set);
            h->cstage_x0_ok = true;
            CHK(time_end(h));
            if (rc != 0) return fail(GBP_EHIP, "general sweep launch failed: %s", hipGetErrorString((hipError_t)rc));
        }
        if (!defer_big) CHK(launch_finish_parts(h, h->stream));
        if (h->p.C) {
            // One workgroup per camera.  Short runs (a camera with a few hundred factors: graphs with thousands of cameras) leave most
            // of a 256-thread block idle through its reduction and 6x6 solve: 128 threads do 1M factors x 2 000 / 3 000 cameras in
            // 125.6 / 127.4 us per sweep against 134.5 / 143.7, 3M factors x 13 682 cameras in 439 against 509; from ~700 factors per
            // camera on the two are equal, at 2 000 per camera 256 threads win (122.9 against 128.0).  The block size fixes the order of
            // the sums, so it depends on the graph's shape alone (GBP_CAM_BLOCK overrides, experiments).
            static const int forced = getenv("GBP_CAM_BLOCK") ? atoi(getenv("GBP_CAM_BLOCK")) : 0;
            // (and one wave per camera below 200 factors per camera: 1M factors x 20 000 cameras 176 against 243 us, 200k x 5 000 50.5 against 65.8)
            const int cam_block = forced ? forced : ((long long)h->p.F < 200LL * h->p.C ? 64 : (long long)h->p.F < 640LL * h->p.C ? 128 : BLOCK);
            if (merged && peer) {
                // peer-store exchange: sum -> push -> wait -> finish in this one launch; the grid must be resident at once (its workgroups
                // wait for other ranks' workgroups), so never more workgroups than the occupancy query admits on this device
                if (!h->staged_xchg_blocks[cam_block >> 7]) {
                    int per_cu = 0;
                    const void *fn = cam_block == 64 ? reinterpret_cast<const void *>(&k_cam_staged_xchg<64>)
                                   : cam_block == 128 ? reinterpret_cast<const void *>(&k_cam_staged_xchg<128>) : reinterpret_cast<const void *>(&k_cam_staged_xchg<BLOCK>);
                    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, cam_block, 0));
                    if (per_cu < 1 || h->n_cus < 1) return fail(GBP_EHIP, "occupancy query of the staged exchange kernel failed");
                    int xb = per_cu * h->n_cus;
                    if (const char *e = getenv("GBP_XCHG_BLOCKS")) xb = std::max(1, std::min(xb, atoi(e)));
                    h->staged_xchg_blocks[cam_block >> 7] = xb;
                }
                const dim3 grid(std::min(h->p.C, h->staged_xchg_blocks[cam_block >> 7]));
                PeerWait w = *merged;
                w.clk = h->clk_cur ? h->clk_cur + 2 : nullptr;
                if (cam_block == 64) hipLaunchKernelGGL(k_cam_staged_xchg<64>, grid, dim3(64), 0, h->stream, h->p, partial, *peer, w);
                else if (cam_block == 128) hipLaunchKernelGGL(k_cam_staged_xchg<128>, grid, dim3(128), 0, h->stream, h->p, partial, *peer, w);
                else hipLaunchKernelGGL(k_cam_staged_xchg<BLOCK>, grid, dim3(BLOCK), 0, h->stream, h->p, partial, *peer, w);
                HIPCHK(hipGetLastError());
                if (finished) *finished = true;
                return GBP_OK;
            }
            if (cam_block == 64) hipLaunchKernelGGL(k_cam_partial_staged<64>, dim3(h->p.C), dim3(64), 0, h->stream, h->p, partial, finish);
            else if (cam_block == 128) hipLaunchKernelGGL(k_cam_partial_staged<128>, dim3(h->p.C), dim3(128), 0, h->stream, h->p, partial, finish);
            else hipLaunchKernelGGL(k_cam_partial_staged<BLOCK>, dim3(h->p.C), dim3(BLOCK), 0, h->stream, h->p, partial, finish);
        }
        HIPCHK(hipGetLastError());
        if (peer) CHK(launch_peer_push(h, partial, *peer));
        if (finished) *finished = finish != 0 && h->p.C > 0;
        return GBP_OK;
    }
    CHK(launch_lmk_beliefs(h));                 // update_all_beliefs: from the stored messages
    CHK(launch_cam_partial(h, partial));
    if (peer) CHK(launch_peer_push(h, partial, *peer));
    return GBP_OK;
}


int gbp::launch_peer_selftest(gbp_ba *h, const PeerOut &po, double *mine, int rank, long long ticks, int *d_out)
{
    hipLaunchKernelGGL(k_peer_selftest, dim3(1), dim3(64), 0, h->stream, po, mine, rank, ticks, d_out);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

// -------------------------------------------------------------------------------- sweep ---

// ---- the dense message remainder on demand -------------------------------------------------------------------------------
// A message is stored as coefficients in the rows of its factor's Jacobian (gbp_math.hpp).  The one thing that does not fit is a
// factor that is DAMPED in the message computation that moves its linearisation point: d * (old eta) lies in the span of the OLD
// Jacobian.  The reference allows it at any time (compute_all_factors with damping on, gbp.py:60-62; relinearise_factors followed by
// compute_all_messages(local_relin=False), gbp.py:46-54); graphs created with num_undamped_iters = 0 carry the out-of-span part
// from the start (Params::xtra, 9 doubles per factor), every other graph gets it HERE, the first time such a call sequence
// shows up, and runs the general sweep (the kernels with the XTRA template flag) until every remainder has decayed to exactly
// zero again -- which the next undamped message of a factor does (x' = d x), i.e. after the next relinearisation wave.
int gbp::enable_remainder(gbp_ba *h)
{
    Params &p = h->p;
    if (p.xtra) return GBP_OK;
    const size_t n = n_slots(h) * XTRA_ROW;
    if (!h->xtra_buf) {
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->xtra_buf), n * sizeof(double)));
        h->allocs.push_back(h->xtra_buf);
    }
    HIPCHK(hipMemsetAsync(h->xtra_buf, 0, n * sizeof(double), h->stream));
    p.xtra = h->xtra_buf;
    p.crow = CSTAGE_ROW;                                     // staged rows carry the remainder too: ensure_staging widens the buffer if need be
    h->lazy_xtra = true; h->lazy_since = 0;
    h->fused_suspended = h->fused.enabled;
    h->fused.enabled = false;
    h->dominant = "k_factor_tile";
    return GBP_OK;
}

// before a message computation: does a pending relinearisation meet a non-zero damping?  (only after stage-wise calls, state loads)
int gbp::remainder_guard(gbp_ba *h, int local_relin, int no_test)
{
    if (!h->pending_possible || h->p.xtra || h->p.eta_damping == 0.0 || !h->p.T) return GBP_OK;
    HIPCHK(hipMemsetAsync(h->d_count, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_count_pending_damped, dim3(grid_for(n_slots(h))), dim3(BLOCK), 0, h->stream, h->p, local_relin, no_test, h->d_count);
    HIPCHK(hipGetLastError());
    int c = 0;
    HIPCHK(hipMemcpyAsync(&c, h->d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (c) CHK(enable_remainder(h));
    return GBP_OK;
}

// after a sweep on a remainder that was switched on on demand: back to the fused sweep once nothing is left of it
int gbp::remainder_release(gbp_ba *h)
{
    if (!h->lazy_xtra || !h->p.xtra || (++h->lazy_since & 15) != 0) return GBP_OK;
    const size_t n = n_slots(h) * XTRA_ROW;
    HIPCHK(hipMemsetAsync(h->d_count, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_count_nonzero, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, h->p.xtra, n, h->d_count);
    HIPCHK(hipGetLastError());
    int c = 0;
    HIPCHK(hipMemcpyAsync(&c, h->d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (c) return GBP_OK;
    h->p.xtra = nullptr;                                     // (the buffer stays for the next time)
    h->p.crow = CSTAGE_PLAIN;                                // the staged rows are 16 doubles wide again (the buffer keeps its size)
    h->lazy_xtra = false;
    if (h->fused_suspended) { h->fused.enabled = true; h->dominant = "k_sweep_fused"; }
    return GBP_OK;
}

// a handle whose remainder was switched on on demand goes back to the state "no remainder" (a checkpoint without one is restored)
int gbp::remainder_drop(gbp_ba *h)
{
    if (!h->lazy_xtra || !h->p.xtra) return GBP_OK;
    h->p.xtra = nullptr;
    h->p.crow = CSTAGE_PLAIN;
    h->lazy_xtra = false;
    if (h->fused_suspended) { h->fused.enabled = true; h->dominant = "k_sweep_fused"; }
    return GBP_OK;
}

extern "C" {

int gbp_ba_update_beliefs(gbp_ba_t *h)
{
    ENTER(h);
    h->resid_ok = false;
    CHK(sweep_begin(h, 0, 0, 0, h->d_partial));
    CHK(launch_cam_finish(h, h->d_partial, 1, 0));
    h->has_beliefs = true;
    return GBP_OK;
}

int gbp_ba_iterate(gbp_ba_t *h, int32_t n_iters, int32_t robustify, int32_t local_relin)
{
    ENTER(h);
    h->resid_ok = false;
    if (n_iters < 0) return fail(GBP_EINVAL, "n_iters < 0");
    for (int it = 0; it < n_iters; ++it) {
        CHK(remainder_guard(h, local_relin, 0));
        bool finished = false;
        CHK(sweep_begin(h, 1, robustify, local_relin, h->d_partial, 1, &finished));
        if (!finished) CHK(launch_cam_finish(h, h->d_partial, 1, 0));
        h->pending_possible = false;                         // every pending relinearisation has been applied
        CHK(remainder_release(h));
    }
    h->has_beliefs = true;
    return GBP_OK;
}

// ---- the reference's stage-wise entry points (gbp.py:46-84) ------------------------------------------------------------
// synchronous_iteration is these four in a row (gbp.py:86-92) and runs as one fused kernel; called one by one they run as stage
// kernels on the same state.  A relinearisation decided by gbp_ba_relinearise / gbp_ba_compute_factors is applied when the
// messages are next computed (gbp_kernels.hpp, state word header).

int gbp_ba_robustify(gbp_ba_t *h)
{
    ENTER(h);
    h->resid_ok = false;
    const Params &p = h->p;
    if (!p.T || p.loss == GBP_LOSS_NONE) return GBP_OK;            // loss None: adaptive variance = gauss_noise_var, nothing stored (gbp.py:302-303)
    const int nb = grid_for(n_slots(h));
    if (p.loss == GBP_LOSS_HUBER) hipLaunchKernelGGL(k_stage_robustify<1>, dim3(nb), dim3(BLOCK), 0, h->stream, p);
    else hipLaunchKernelGGL(k_stage_robustify<2>, dim3(nb), dim3(BLOCK), 0, h->stream, p);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

int gbp_ba_relinearise(gbp_ba_t *h)
{
    ENTER(h);
    h->resid_ok = false;
    if (!h->has_beliefs) return fail(GBP_ESTATE, "relinearise_factors needs beliefs (call update_all_beliefs first; the reference inverts zero matrices here, gbp.py:73)");
    clock_tick(h, true);
    if (h->p.T) hipLaunchKernelGGL(k_stage_relinearise, dim3(grid_for(n_slots(h))), dim3(BLOCK), 0, h->stream, h->p, 0);
    HIPCHK(hipGetLastError());
    h->pending_possible = true;
    return GBP_OK;
}

int gbp_ba_compute_factors(gbp_ba_t *h)
{
    ENTER(h);
    h->resid_ok = false;
    if (!h->has_beliefs) return fail(GBP_ESTATE, "compute_all_factors linearises at the belief means: call update_all_beliefs first");
    // (a factor that is damped when it moves leaves the span its message coefficients live in: the message computation that applies
    //  the move checks for that and switches the dense remainder on, remainder_guard)
    if (h->p.T) hipLaunchKernelGGL(k_stage_relinearise, dim3(grid_for(n_slots(h))), dim3(BLOCK), 0, h->stream, h->p, 1);
    HIPCHK(hipGetLastError());
    h->pending_possible = true;
    return GBP_OK;
}

int gbp_ba_compute_messages(gbp_ba_t *h, int32_t local_relin)
{
    ENTER(h);
    h->resid_ok = false;
    if (!h->has_beliefs) return fail(GBP_ESTATE, "compute_all_messages needs beliefs (call update_all_beliefs first)");
    CHK(remainder_guard(h, local_relin, 1));
    const int slot = (int)(h->sweep_count % RELIN_RING);
    if (slot % (RELIN_RING / 2) == 0)
        HIPCHK(hipMemsetAsync(h->d_relin_ring + (size_t)slot * RELIN_LANES, 0, sizeof(int) * (RELIN_RING / 2) * RELIN_LANES, h->stream));
    h->p.relin_slot = h->d_relin_ring + (size_t)slot * RELIN_LANES;
    h->sweep_count++;
    h->p.stage = STAGE_NO_TEST | STAGE_NO_BELIEFS;
    clock_tick(h, false);                                    // (no relinearisation test in this call: nobody ages)
    h->p.reverse_walk = 0;
    const int rc = launch_factor_stage(h, 0, local_relin);
    h->p.stage = 0;
    h->pending_possible = false;
    return rc;
}

int gbp_ba_shard_begin(gbp_ba_t *h, int32_t with_messages, int32_t robustify, int32_t local_relin, double *partial_dev)
{
    ENTER(h);
    h->resid_ok = false;
    if (!partial_dev) return fail(GBP_EINVAL, "null partial buffer");
    if (with_messages) CHK(remainder_guard(h, local_relin, 0));
    CHK(sweep_begin(h, with_messages, robustify, local_relin, partial_dev));
    if (with_messages) { h->pending_possible = false; CHK(remainder_release(h)); }
    return GBP_OK;
}

int gbp_ba_shard_end(gbp_ba_t *h, const double *gathered_dev, int32_t n_ranks)
{
    ENTER(h);
    h->resid_ok = false;
    if (!gathered_dev || n_ranks < 1) return fail(GBP_EINVAL, "bad gathered buffer / rank count");
    CHK(launch_cam_finish(h, gathered_dev, n_ranks, (size_t)h->p.C * 27));
    h->has_beliefs = true;
    return GBP_OK;
}

// --------------------------------------------------------------------------- diagnostics ---

int gbp_ba_residual_sums(gbp_ba_t *h, double out[2])
{
    ENTER(h);
    CHK(peer_check(h, false));
    if (!out) return fail(GBP_EINVAL, "null argument");
    const Params &p = h->p;
    out[0] = out[1] = 0.0;
    if (!p.F) return GBP_OK;
    if (!h->resid_ok) {                                 // are() then energy() on the same state: one kernel, one round trip
        const int nb = grid_for(n_slots(h));
        hipLaunchKernelGGL(k_residual, dim3(nb), dim3(BLOCK), 0, h->stream, p, h->d_red);
        HIPCHK(hipGetLastError());
        std::vector<double> part;
        CHK(download(h, part, h->d_red, 2 * (size_t)nb));
        h->resid[0] = h->resid[1] = 0.0;
        for (int b = 0; b < nb; ++b) { h->resid[0] += part[2 * b]; h->resid[1] += part[2 * b + 1]; }
        h->resid_ok = true;
    }
    out[0] = h->resid[0]; out[1] = h->resid[1];
    return GBP_OK;
}

int gbp_ba_are(gbp_ba_t *h, double *out)
{
    if (!out) return fail(GBP_EINVAL, "null argument");
    double s[2];
    CHK(gbp_ba_residual_sums(h, s));
    *out = s[0] / (double)h->p.F;               // divides by len(self.factors)  gbp_ba.py:69
    return GBP_OK;
}

int gbp_ba_energy(gbp_ba_t *h, double *out)
{
    if (!out) return fail(GBP_EINVAL, "null argument");
    double s[2];
    CHK(gbp_ba_residual_sums(h, s));
    *out = s[1];
    return GBP_OK;
}

// ------------------------------------------------------------------------ instrumentation ---

int gbp_ba_set_kernel_timing(gbp_ba_t *h, int32_t enable)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->timing = enable != 0;
    h->timing_every = enable > 1 ? enable : 1;
    h->timing_tick = 0;
    h->ev_used = 0;
    h->clk_used = 0; h->clk_cur = nullptr;
    if (enable) {
        if (!h->d_clk) {
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->d_clk), sizeof(unsigned long long) * 6 * CLK_RING));
            h->allocs.push_back(h->d_clk);
            HIPCHK(hipDeviceGetAttribute(&h->clk_rate_khz, hipDeviceAttributeWallClockRate, h->device));
        }
        if (!h->clk_calibrated) {
            // The rate of wall_clock64: hipDeviceAttributeWallClockRate says 100 MHz, and on some boxes of the pool the counter runs
            // ~7 % faster than that (stamped kernel times came out longer than the step that contains them, while HIP events and the
            // wall clock agreed with each other).  Measured once per handle: two stamps 20 ms apart against the HIP events around them.
            unsigned long long *d_cal = nullptr, cal[2] = {0, 0};
            hipEvent_t e0, e1;
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_cal), 2 * sizeof(unsigned long long)));
            HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k_clk_stamp, dim3(1), dim3(64), 0, h->stream, d_cal);
            HIPCHK(hipEventRecord(e0, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            hipLaunchKernelGGL(k_clk_stamp, dim3(1), dim3(64), 0, h->stream, d_cal + 1);
            HIPCHK(hipEventRecord(e1, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            HIPCHK(hipMemcpy(cal, d_cal, sizeof cal, hipMemcpyDeviceToHost));
            (void)hipFree(d_cal); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            if (ms > 1.f && cal[1] > cal[0]) {
                const double khz = (double)(cal[1] - cal[0]) / (double)ms;
                if (khz > 0.5 * h->clk_rate_khz && khz < 2.0 * h->clk_rate_khz) h->clk_rate_khz_measured = khz;
            }
            h->clk_calibrated = true;
        }
        hipLaunchKernelGGL(k_clk_init, dim3((6 * CLK_RING + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, h->stream, h->d_clk, 6 * CLK_RING);
        HIPCHK(hipGetLastError());
    }
    return GBP_OK;
}

int gbp_ba_get_sweep_clocks(gbp_ba_t *h, double *us6, int32_t cap, int32_t *n_sweeps)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    const int n = std::min(h->clk_used, CLK_RING);
    if (n_sweeps) *n_sweeps = n;
    if (!us6 || !n || cap <= 0) return GBP_OK;
    std::vector<unsigned long long> raw;
    CHK(download(h, raw, h->d_clk, 6 * (size_t)n));
    unsigned long long t0 = ~0ull;
    for (unsigned long long v : raw) if (v != 0ull && v != ~0ull) t0 = std::min(t0, v);
    const double khz = h->clk_rate_khz_measured > 0.0 ? h->clk_rate_khz_measured : (double)h->clk_rate_khz;
    const double us_per_tick = khz > 0.0 ? 1e3 / khz : 0.01;
    for (int i = 0; i < n && i < cap; ++i)
        for (int k = 0; k < 6; ++k) {
            const unsigned long long v = raw[6 * (size_t)i + k];
            us6[6 * (size_t)i + k] = (v == 0ull || v == ~0ull) ? NAN : (double)(v - t0) * us_per_tick;
        }
    return GBP_OK;
}

int gbp_ba_get_kernel_timing(gbp_ba_t *h, double *total_ms, int32_t *n_launches, const char **kernel_name)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    double tot = 0.0;
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (n_launches) *n_launches = (int32_t)(h->ev_used / 2);
    if (kernel_name) *kernel_name = h->dominant;
    h->ev_used = 0;
    return GBP_OK;
}

int gbp_ba_get_kernel_times(gbp_ba_t *h, double *ms, int32_t cap, int32_t *n_launches)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    const int32_t n = (int32_t)(h->ev_used / 2);
    if (n_launches) *n_launches = n;
    for (int32_t i = 0; i < n && i < cap && ms; ++i) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, h->ev[2 * (size_t)i], h->ev[2 * (size_t)i + 1]));
        ms[i] = t;
    }
    return GBP_OK;
}

// phase profile of the last fused sweep (GBP_PHASE_TIMING builds; rows = workgroups x waves, NPHASE columns of s_memtime ticks)
int gbp_ba_phase_profile(gbp_ba_t *h, uint64_t *out, int32_t cap_rows, int32_t *n_rows, int32_t *n_cols)
{
    ENTER(h);
    if (n_rows) *n_rows = 0;
    if (n_cols) *n_cols = NPHASE;
#ifdef GBP_PHASE_TIMING
    if (!h->fused.enabled || !h->fused.args.phase) return fail(GBP_ESTATE, "no phase profile: the fused sweep is not in use");
    const int rows = h->fused.n_blocks * WAT_WAVES;
    if (n_rows) *n_rows = rows;
    if (out && cap_rows >= rows) {
        HIPCHK(hipMemcpyAsync(out, h->fused.args.phase, sizeof(uint64_t) * (size_t)rows * NPHASE, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return GBP_OK;
#else
    (void)out; (void)cap_rows;
    return fail(GBP_ESTATE, "not a GBP_PHASE_TIMING build (tools/phase_profile.py compiles one)");
#endif
}


}  // extern "C"
