// gbp_capi.hip -- libgbp_hip.so, life cycle: error strings, create / destroy (the graph build runs on the device: gbp_build.hpp),
// streams, priors, the BAL reader, what the plan decided, layout checks.  The other translation units: gbp_handle.hpp.
//
// No CPU compute path lives in this library: every sweep, belief update and diagnostic is a HIP kernel, and so is the graph build
// (ordering, tile packing, slot assignment, initial linearisation points, prior maxima).  Host code only (a) sizes the allocations
// from the tile count the device reports, (b) packs / unpacks symmetric matrices for the views, (c) owns handles, streams and the
// optional RCCL communicator.
#include "gbp_handle.hpp"
#include "gbp_build.hpp"
#include "gbp_balio.hpp"

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <mutex>
#include <new>

static thread_local std::string g_err;

namespace gbp {
int set_error(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace gbp

extern "C" {


int gbp_abi_version(void) { return GBP_ABI_VERSION; }
const char *gbp_last_error(void) { return g_err.c_str(); }

void gbp_ba_destroy(gbp_ba_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *ptr : h->allocs) (void)hipFree(ptr);
    if (h->d_tmp) (void)hipFree(h->d_tmp);
    for (void *q : h->snap) if (q) (void)hipFree(q);
    if (h->d_send) (void)hipFree(h->d_send);
    if (h->d_recv) (void)hipFree(h->d_recv);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    for (int i = 0; i < 2; ++i) { if (h->h_mu[i]) (void)hipHostFree(h->h_mu[i]); if (h->ev_landed[i]) (void)hipEventDestroy(h->ev_landed[i]); }
    if (h->ev_packed) (void)hipEventDestroy(h->ev_packed);
    shard_comm_release(h);
    peer_release(h);
    if (h->peer.mailbox) (void)hipFree(h->peer.mailbox);
    if (h->peer.d_ctl) (void)hipFree(h->peer.d_ctl);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    fused_destroy(h->fused);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

extern "C++" {
// host -> device copy of a caller array that is released at the end of create (or the caller's own device pointer)
template <typename T>
static int stage_input(gbp_ba *h, const T *src, size_t n, bool on_device, std::vector<void *> &scratch, const T **out)
{
    if (on_device || !n) { *out = src; return GBP_OK; }
    void *q = nullptr;
    HIPCHK(hipMallocAsync(&q, n * sizeof(T), h->stream));
    scratch.push_back(q);
    HIPCHK(hipMemcpyAsync(q, src, n * sizeof(T), hipMemcpyHostToDevice, h->stream));
    *out = static_cast<const T *>(q);
    return GBP_OK;
}

// buffers only the build needs come from the stream-ordered pool (no device synchronisation per allocation or release)
template <typename T>
static int scratch_alloc(gbp_ba *h, std::vector<void *> &scratch, T **out, size_t n)
{
    void *q = nullptr;
    HIPCHK(hipMallocAsync(&q, std::max<size_t>(n, 1) * sizeof(T), h->stream));
    scratch.push_back(q);
    *out = static_cast<T *>(q);
    return GBP_OK;
}

}  // extern "C++"

// GBP_BUILD_TIMING=1: wall time of the stages of gbp_ba_create on stderr (each mark synchronises the stream: diagnostic only)
struct BuildClock {
    bool on; hipStream_t s; std::chrono::steady_clock::time_point t0;
    BuildClock(hipStream_t st) : on(getenv("GBP_BUILD_TIMING") != nullptr), s(st), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what)
    {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[gbp build] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
        t0 = t1;
    }
};

static int build_graph(gbp_ba *h, const gbp_ba_desc_t *d, std::vector<void *> &scratch, int n_cus)
{
    BuildClock clk(h->stream);
    const int C = d->n_cams, L = d->n_lmks, F = d->n_factors;
    Params &p = h->p;
    const bool dev_in = (d->flags & GBP_FLAG_DEVICE_INPUT) != 0;
    const size_t Fz = (size_t)F;
    const int *cam_idx = nullptr, *lmk_idx = nullptr;
    const double *meas = nullptr, *cam_means = nullptr, *lmk_means = nullptr;
    CHK(stage_input(h, d->cam_idx, Fz, dev_in, scratch, &cam_idx)); CHK(stage_input(h, d->lmk_idx, Fz, dev_in, scratch, &lmk_idx));
    CHK(stage_input(h, d->meas, Fz * 2, dev_in, scratch, &meas));
    CHK(stage_input(h, d->cam_means, (size_t)C * 6, dev_in, scratch, &cam_means));
    CHK(stage_input(h, d->lmk_means, (size_t)L * 3, dev_in, scratch, &lmk_means));

    clk.mark("stage inputs");
    // 1. ids in range?  already camera-major?
    int *d_flags = nullptr;
    CHK(scratch_alloc(h, scratch, &d_flags, 2));
    HIPCHK(hipMemsetAsync(d_flags, 0, 2 * sizeof(int), h->stream));
    if (F) hipLaunchKernelGGL(k_check_ids, dim3(grid_for(Fz)), dim3(BLOCK), 0, h->stream, cam_idx, lmk_idx, F, C, L, d_flags);
    HIPCHK(hipGetLastError());
    int flags[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (flags[0]) return fail(GBP_EINVAL, "an observation references a camera outside [0,%d) or a landmark outside [0,%d)", C, L);

    clk.mark("check ids");
    // 2. reference order: camera-major, stable in file order (gbp_ba.py:128-130)
    int *iota = nullptr, *ref_file = nullptr, *lm_key = nullptr, *lm2ref = nullptr, *lptr = nullptr, *cptr = nullptr, *cadj = nullptr, *cpos = nullptr;
    CHK(dev_alloc(h, &h->d_ref_cam, std::max<size_t>(Fz, 1), false)); CHK(dev_alloc(h, &h->d_ref_lmk, std::max<size_t>(Fz, 1), false));
    CHK(scratch_alloc(h, scratch, &iota, Fz)); CHK(scratch_alloc(h, scratch, &lm_key, Fz)); CHK(scratch_alloc(h, scratch, &lm2ref, Fz));
    CHK(scratch_alloc(h, scratch, &lptr, (size_t)L + 1));
    CHK(dev_alloc(h, &cptr, (size_t)C + 1)); CHK(dev_alloc(h, &cadj, std::max<size_t>(Fz, 1)));
    int bits_c = 1, bits_l = 1;
    while ((1 << bits_c) < C) ++bits_c;
    while ((1 << bits_l) < L) ++bits_l;
    void *sort_tmp = nullptr;
    const size_t sort_bytes = F ? std::max(sort_pairs_tmp_bytes(Fz, bits_c), sort_pairs_tmp_bytes(Fz, bits_l)) : 0;
    if (sort_bytes) { HIPCHK(hipMallocAsync(&sort_tmp, sort_bytes, h->stream)); scratch.push_back(sort_tmp); }
    if (F) hipLaunchKernelGGL(k_iota, dim3(grid_for(Fz)), dim3(BLOCK), 0, h->stream, iota, F);
    const bool sorted = !flags[1];
    if (F && !sorted) {
        CHK(scratch_alloc(h, scratch, &ref_file, Fz));
        HIPCHK((hipError_t)sort_pairs(sort_tmp, sort_bytes, cam_idx, h->d_ref_cam, iota, ref_file, Fz, bits_c, h->stream));
        hipLaunchKernelGGL(k_gather_int, dim3(grid_for(Fz)), dim3(BLOCK), 0, h->stream, lmk_idx, ref_file, h->d_ref_lmk, F);
    } else if (F) {
        HIPCHK(hipMemcpyAsync(h->d_ref_cam, cam_idx, Fz * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_ref_lmk, lmk_idx, Fz * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    }
    hipLaunchKernelGGL(k_lower_bounds, dim3(grid_for((size_t)C + 1)), dim3(BLOCK), 0, h->stream, h->d_ref_cam, F, cptr, C);
    // 3. landmark-major, stable in reference id (= VariableNode.adj_factors order, gbp_ba.py:139)
    if (F) HIPCHK((hipError_t)sort_pairs(sort_tmp, sort_bytes, h->d_ref_lmk, lm_key, iota, lm2ref, Fz, bits_l, h->stream));
    hipLaunchKernelGGL(k_lower_bounds, dim3(grid_for((size_t)L + 1)), dim3(BLOCK), 0, h->stream, lm_key, F, lptr, L);
    HIPCHK(hipGetLastError());

    clk.mark("orders (allocs + 2 sorts)");
    // 4. tiles: up to 64 slots / TILE_LMKS whole landmarks each; over-sized landmarks become chunk tiles (a piece of one landmark
    //    each).  Next-fit packing as list ranking on the device (gbp_build.hpp); only the tile count, the number of over-sized
    //    landmarks and the smallest landmark degree come back -- the last one decides whether the dense packing may replace this one.
    const int LP = L + 1;
    const int n_pblocks = std::max(1, (L + PACK_BLOCK - 1) / PACK_BLOCK), n_nodes = n_pblocks * TILE_LMKS + 1;   // (block, entry) nodes + the end
    int levels = 1;
    while ((1 << levels) < n_pblocks + 1) ++levels;
    int *nxt = nullptr, *w0 = nullptr, *jump = nullptr, *wsum = nullptr, *bpos = nullptr, *pos = nullptr, *d_lrow0 = nullptr, *d_lrow1 = nullptr,
        *d_big_list = nullptr, *d_cnt = nullptr;
    CHK(scratch_alloc(h, scratch, &nxt, (size_t)LP)); CHK(scratch_alloc(h, scratch, &w0, (size_t)LP));
    CHK(scratch_alloc(h, scratch, &jump, (size_t)levels * n_nodes)); CHK(scratch_alloc(h, scratch, &wsum, (size_t)levels * n_nodes));
    CHK(scratch_alloc(h, scratch, &bpos, (size_t)n_nodes)); CHK(scratch_alloc(h, scratch, &pos, (size_t)LP));
    CHK(scratch_alloc(h, scratch, &d_lrow0, (size_t)L)); CHK(scratch_alloc(h, scratch, &d_lrow1, (size_t)L));
    CHK(scratch_alloc(h, scratch, &d_big_list, (size_t)L)); CHK(scratch_alloc(h, scratch, &d_cnt, 1));
    HIPCHK(hipMemsetAsync(pos, 0xff, sizeof(int) * (size_t)LP, h->stream));
    HIPCHK(hipMemsetAsync(bpos, 0xff, sizeof(int) * (size_t)n_nodes, h->stream));
    HIPCHK(hipMemsetAsync(bpos, 0, sizeof(int), h->stream));                      // the chain enters block 0 at landmark 0 with tile 0
    HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_pack_next, dim3(grid_for((size_t)LP)), dim3(BLOCK), 0, h->stream, lptr, L, nxt, w0);
    hipLaunchKernelGGL(k_pack_block_walk, dim3(grid_for((size_t)n_nodes)), dim3(BLOCK), 0, h->stream, nxt, w0, L, n_pblocks, jump, wsum);
    for (int k = 0; k + 1 < levels; ++k)
        hipLaunchKernelGGL(k_pack_double, dim3(grid_for((size_t)n_nodes)), dim3(BLOCK), 0, h->stream, jump + (size_t)k * n_nodes, wsum + (size_t)k * n_nodes,
                           jump + (size_t)(k + 1) * n_nodes, wsum + (size_t)(k + 1) * n_nodes, n_nodes);
    for (int k = levels - 1; k >= 0; --k)
        hipLaunchKernelGGL(k_pack_mark, dim3(grid_for((size_t)n_nodes)), dim3(BLOCK), 0, h->stream, jump + (size_t)k * n_nodes, wsum + (size_t)k * n_nodes, bpos,
                           n_nodes);
    hipLaunchKernelGGL(k_pack_block_fill, dim3(grid_for((size_t)n_pblocks)), dim3(BLOCK), 0, h->stream, nxt, w0, L, n_pblocks, bpos, pos);
    HIPCHK(hipGetLastError());
    int T = 0, min_deg = 0x7fffffff, *d_mindeg = nullptr;
    CHK(scratch_alloc(h, scratch, &d_mindeg, 1));
    HIPCHK(hipMemcpyAsync(d_mindeg, &min_deg, sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (L) hipLaunchKernelGGL(k_min_degree, dim3(grid_for((size_t)L)), dim3(BLOCK), 0, h->stream, lptr, L, d_mindeg);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&T, pos + L, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&min_deg, d_mindeg, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (L == 0) T = 0;
    // Whole landmarks per tile (T tiles from the list ranking above), or -- when that would leave more than 15 % of the slots empty
    // and every landmark has at least three factors -- the dense packing: tile t = factors [64 t, 64 t + 64), landmarks may span tiles
    // (gbp_build.hpp).  One million factors at 40 per landmark: 25 000 tiles of 40 -> 15 625 full ones.  GBP_PACK=whole|dense overrides
    // (tests, A/B runs); "dense" still needs the three factors per landmark.
    bool dense = false;
    if (T > 0 && F > 0 && min_deg >= 3) {
        const char *e = getenv("GBP_PACK");
        dense = e ? strcmp(e, "dense") == 0 : (double)F < 0.85 * 64.0 * (double)T;
    }
    if (dense) T = (F + WTILE - 1) / WTILE;
    if (T < 0 || (int64_t)T * WTILE > INT32_MAX) return fail(GBP_EINVAL, "the graph needs %d tiles: slot indices would not fit 32 bits", T);
    const size_t S = std::max<size_t>((size_t)T * WTILE, 1);
    p.T = T;
    int4 *d_tiles = nullptr;
    CHK(dev_alloc(h, &d_tiles, std::max<size_t>((size_t)T, 1)));
    int n_big = 0;
    if (dense) {
        hipLaunchKernelGGL(k_dense_tiles, dim3(grid_for((size_t)T)), dim3(BLOCK), 0, h->stream, lptr, L, F, T, d_tiles);
        hipLaunchKernelGGL(k_dense_rows, dim3(grid_for((size_t)L)), dim3(BLOCK), 0, h->stream, lptr, L, d_lrow0, d_lrow1);
        HIPCHK(hipGetLastError());
    } else {
        if (L) hipLaunchKernelGGL(k_pack_emit, dim3(grid_for((size_t)L)), dim3(BLOCK), 0, h->stream, lptr, L, nxt, pos, d_tiles, d_lrow0, d_lrow1,
                                  d_big_list, d_cnt);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&n_big, d_cnt, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    // 0: every landmark inside one tile; 1: whole landmarks, those above 64 factors in chunk tiles; 2: dense.  From 1 on some landmarks span
    // tiles: the tiles write partial sums (Params::parts) and k_lmk_finish_parts forms those beliefs after every sweep.
    h->pack_mode = dense ? 2 : (n_big ? 1 : 0);
    h->n_big = n_big;

    clk.mark("tile packing");
    // 5. per-slot data
    bool general_sweep = false;
    {
        int n_wg = std::max(1, std::min(T, n_cus));
        if (const char *nb = getenv("GBP_FUSED_BLOCKS")) n_wg = std::max(1, std::min(n_wg, atoi(nb)));      // (experiment switch, as in fused_plan)
        const int cgmax = fused_max_cams();      // (host arithmetic on the LDS budget: gbp_fused_plan.hpp)
        // Camera WINDOWS.  Workgroup b of the fused sweep walks the tiles [b T / n, (b + 1) T / n) and needs table rows for the cameras
        // of THOSE tiles only.  In a sequence -- landmarks numbered along the trajectory, each seen from neighbouring cameras, now and
        // then from a place visited before -- that is a few dozen cameras however many the graph has: the table becomes [set][27] per
        // workgroup (k_wg_cam_sets: the distinct cameras of its tiles; a 16-bit map over the interval they lie in turns a camera into
        // its table row), more cameras than fit the LDS as a whole still run the fused sweep, and the tables written and reduced every
        // sweep shrink from workgroups x cameras rows to the sum of the sets.  Taken when the whole table would not fit, or when the sets
        // add up to at most 0.7 of it; GBP_WINDOWS=0 / 1: never / whenever they fit.
        h->wg_win.clear(); h->wg_cams.clear();
        long long rows = (long long)n_wg * C;
        if (T > 0 && C > 0 && !(h->flags & GBP_FLAG_NO_FUSED) && p.num_undamped != 0) {
            int2 *d_rng = nullptr;
            int *d_lists = nullptr, *d_counts = nullptr;
            const int cap = cgmax;
            CHK(scratch_alloc(h, scratch, &d_rng, (size_t)n_wg));
            CHK(scratch_alloc(h, scratch, &d_lists, (size_t)n_wg * cap)); CHK(scratch_alloc(h, scratch, &d_counts, (size_t)n_wg));
            hipLaunchKernelGGL(k_wg_cam_range, dim3((n_wg + BLOCK / 64 - 1) / (BLOCK / 64)), dim3(BLOCK), 0, h->stream, d_tiles, d_lrow0, lptr, lm2ref,
                               h->d_ref_cam, T, n_wg, d_rng);
            HIPCHK(hipGetLastError());
            std::vector<int2> rng((size_t)n_wg);
            HIPCHK(hipMemcpyAsync(rng.data(), d_rng, sizeof(int2) * (size_t)n_wg, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            int max_width = 0;
            for (const int2 &r : rng) max_width = std::max(max_width, r.y >= r.x ? r.y - r.x + 1 : 0);
            const size_t bm_bytes = ((size_t)(max_width + 31) / 32 + SETS_THREADS + 1) * sizeof(unsigned);
            if (bm_bytes <= 64 * 1024 && fused_shmem_windows(1, max_width) <= (size_t)LDS_BYTES) {      // (else: intervals no map of the sweep could cover)
                hipLaunchKernelGGL(k_wg_cam_sets, dim3(n_wg), dim3(SETS_THREADS), bm_bytes, h->stream, d_tiles, d_lrow0, lptr, lm2ref, h->d_ref_cam, T, n_wg,
                                   d_rng, cap, d_lists, d_counts);
                HIPCHK(hipGetLastError());
                std::vector<int> counts((size_t)n_wg);
                HIPCHK(hipMemcpyAsync(counts.data(), d_counts, sizeof(int) * (size_t)n_wg, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
                long long sum = 0;
                int max_set = 0;
                for (int n : counts) { sum += n; max_set = std::max(max_set, n); }
                const char *e = getenv("GBP_WINDOWS");
                // (a 125k-factor share of the headline graph, 500 random cameras: sets 0.63 of the whole tables, 26.6 against 29.2 us per
                //  sweep; a 250k share: 0.86, 36.6 against 34.4 -- tools/sparse_probe.sh)
                const bool want = e ? atoi(e) != 0 : (C > cgmax || 10 * sum <= 7 * rows);
                if (want && max_set <= cap && fused_shmem_windows(max_set, max_width) <= (size_t)LDS_BYTES) {
                    std::vector<int> lists((size_t)n_wg * cap);
                    HIPCHK(hipMemcpyAsync(lists.data(), d_lists, sizeof(int) * lists.size(), hipMemcpyDeviceToHost, h->stream));
                    HIPCHK(hipStreamSynchronize(h->stream));
                    h->wg_win.resize((size_t)n_wg); h->wg_cams.reserve((size_t)sum);
                    for (int b = 0; b < n_wg; ++b) {
                        const int n = counts[(size_t)b];
                        h->wg_win[(size_t)b] = make_int4(n ? rng[(size_t)b].x : 0, n, (int)h->wg_cams.size(), n ? rng[(size_t)b].y - rng[(size_t)b].x + 1 : 0);
                        h->wg_cams.insert(h->wg_cams.end(), lists.begin() + (size_t)b * cap, lists.begin() + (size_t)b * cap + n);
                    }
                    rows = sum;
                }
            }
        }
        const bool windowed = !h->wg_win.empty();
        // Few factors per camera: the fused sweep writes (and its reduce reads back) one 224-byte table row per camera and WORKGROUP
        // whatever the graph's size, the staged form one 128-byte row per FACTOR.  Below ~0.75 factors per (workgroup, camera) the
        // staged sweep is the faster one -- 13k / 30k / 60k / 90k factors x 500 cameras: 18.6 / 20.5 / 24.2 / 29.1 against 26.3 /
        // 28.6 / 28.9 / 30.2 us per sweep (profiles/r04_shards.json); round 5: 62.5k 22.1 against 28.6; 125k is a tie (30.6 against 31.4,
        // with the peer-store exchange in the loop 34.5 against 35.3), 250k 44.6 against 36.9 (profiles/r05_shards.json) -- e.g. a rank's
        // share of the headline graph at 16 ranks and beyond.  Not a byte count: fr1desk (13 298 factors, 63 cameras, 221 workgroups: 0.96
        // per pair) runs 13.9 us fused against ~16 staged, so the threshold stays below 1.  GBP_STAGED_BELOW overrides it (0: never).
        double staged_below = 0.75;
        if (const char *e = getenv("GBP_STAGED_BELOW")) staged_below = atof(e);
        const bool sparse = (double)F < staged_below * (double)rows;      // (windows: the rows the tables really have)
        h->staged_auto = sparse && !(h->flags & (GBP_FLAG_FORCE_FUSED | GBP_FLAG_NO_FUSED));
        if (h->staged_auto) h->flags |= GBP_FLAG_NO_FUSED;
        if (h->flags & GBP_FLAG_NO_FUSED) { h->wg_win.clear(); h->wg_cams.clear(); }
        general_sweep = (h->flags & GBP_FLAG_NO_FUSED) || p.num_undamped == 0 || (C > cgmax && !windowed);   // its staging buffer is streamed every sweep too
        const size_t need = (general_sweep ? std::max<size_t>(Fz, 1) * p.crow * sizeof(double) + (64 << 8) : 0) + S * (LIN_ROWS + MSG_ROWS + (p.num_undamped == 0 ? XTRA_ROW : 0) + (p.loss != 0 ? 1 : 0)) * sizeof(double) + S * sizeof(int)
                          + (size_t)std::max(L, 1) * LREC * sizeof(double) + (size_t)std::max<long long>(rows, 1) * (TROW * sizeof(double) + 2 * sizeof(int)) + (size_t)n_wg * sizeof(int4) + (size_t)std::max(C, 1) * sizeof(int2) + 4 * 4096
                          + (size_t)std::max(C, 1) * (CAMREC + CBEL + 27 + 27 + 1) * sizeof(double) + (size_t)std::max(L, 1) * sizeof(double)
                          + 2 * (size_t)grid_for(S) * sizeof(double) + (size_t)RELIN_RING * RELIN_LANES * sizeof(int)
                          + (size_t)(n_wg + 1) * sizeof(int) + (h->pack_mode ? 2 * (size_t)std::max(T, 1) * PART_ROW * sizeof(double) : 0) + (64 << 12);
        CHK(arena_reserve(h, need));
    }
    if (general_sweep && F > 0) { CHK(dev_alloc(h, &p.cstage, Fz * p.crow)); h->cstage_cap = p.crow; }   // out of the same arena (else: on first use, ensure_staging)
    CHK(dev_alloc(h, &p.lin, S * LIN_ROWS)); CHK(dev_alloc(h, &p.msg, S * MSG_ROWS));
    if (p.num_undamped == 0) CHK(dev_alloc(h, &p.xtra, S * XTRA_ROW));      // damped in the relinearising sweep: gbp_math.hpp header
    if (p.loss != 0) CHK(dev_alloc(h, &p.avar, S));         // adaptive variances: robust losses only
    CHK(dev_alloc(h, &cpos, S));
    CHK(dev_alloc(h, &p.lrec, (size_t)std::max(L, 1) * LREC));
    if (h->pack_mode) CHK(dev_alloc(h, &p.parts, 2 * (size_t)std::max(T, 1) * PART_ROW));
    CHK(dev_alloc(h, &p.cbel, (size_t)std::max(C, 1) * CAMREC)); CHK(dev_alloc(h, &p.cprior, (size_t)std::max(C, 1) * 27));
    CHK(dev_alloc(h, &p.cbelief, (size_t)std::max(C, 1) * CBEL));
    p.tiles = d_tiles; p.cptr = cptr; p.cadj = cadj; p.cpos = cpos;
    if (T) {
        BuildArgs a{d_tiles, d_lrow0, lptr, lm2ref, h->d_ref_cam, ref_file, cam_means, lmk_means, meas, cadj, cpos};
        hipLaunchKernelGGL(k_build_tiles, dim3((T + BLOCK / 64 - 1) / (BLOCK / 64)), dim3(BLOCK), 0, h->stream, p, a);
    }
    clk.mark("allocs + k_build_tiles");
    // 6. variables
    if (C + L) hipLaunchKernelGGL(k_init_vars, dim3(grid_for((size_t)C + L)), dim3(BLOCK), 0, h->stream, p, cam_means, lmk_means, d_lrow0, d_lrow1);
    HIPCHK(hipGetLastError());

    CHK(dev_alloc(h, &h->d_partial, (size_t)std::max(C, 1) * 27));
    CHK(dev_alloc(h, &h->d_red, 2 * (size_t)grid_for(S)));
    CHK(dev_alloc(h, &h->d_relin_ring, (size_t)RELIN_RING * RELIN_LANES));
    CHK(dev_alloc(h, &h->d_count, 2));
    CHK(dev_alloc(h, &h->d_varmax, (size_t)std::max(C + L, 1), false));

    clk.mark("vars + small allocs");
    if (getenv("GBP_DEBUG_LAYOUT")) {
        int bad = 0;
        CHK(gbp_ba_check_layout(h, &bad));
        if (bad) return fail(GBP_ESTATE, "internal layout error: %d slots do not decode to their reference factor", bad);
    }
    if (!(h->flags & GBP_FLAG_NO_FUSED) && !p.xtra) {
        HIPCHK(hipStreamSynchronize(h->stream));            // tiles[].w (max rank) is written by k_build_tiles
        h->fused.alloc = arena_take; h->fused.alloc_ctx = h;
        int rc = plan_fused_sweep(h, n_cus);
        if (rc < 0) return fail(GBP_EHIP, "building the fused sweep plan failed (%d)", rc);
        if (h->fused.enabled) h->dominant = "k_sweep_fused";
    }
    HIPCHK(hipStreamSynchronize(h->stream));                // the staged inputs are released by the caller
    clk.mark("fused plan");
    if (getenv("GBP_PRINT_PTRS"))
        fprintf(stderr, "[gbp ptrs] lin %p msg %p lrec %p cbel %p tables %p\n", (void *)p.lin, (void *)p.msg, (void *)p.lrec, (void *)p.cbel,
                (void *)h->fused.args.block_partials);
    return GBP_OK;
}

static int create_impl(gbp_ba *h, const gbp_ba_desc_t *d)
{
    const int C = d->n_cams;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GBP_ENODEV, "no HIP device visible: libgbp_hip.so has no CPU path");
    if (d->device < 0 || d->device >= ndev) return fail(GBP_EINVAL, "device %d out of range (%d visible)", d->device, ndev);
    h->device = d->device;
    HIPCHK(hipSetDevice(h->device));
    // hipGetDeviceProperties costs ~1 ms: asked once per device and process
    static std::mutex prop_mutex;
    static std::vector<std::pair<int, std::string>> prop_cache;       // [device] = {CU count, arch}
    int n_cus = 0;
    std::string arch;
    {
        std::lock_guard<std::mutex> lock(prop_mutex);
        if ((int)prop_cache.size() < ndev) prop_cache.resize(ndev, {0, std::string()});
        if (prop_cache[h->device].first == 0) {
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, h->device));
            prop_cache[h->device] = {prop.multiProcessorCount, std::string(prop.gcnArchName)};
        }
        n_cus = prop_cache[h->device].first; arch = prop_cache[h->device].second;
    }
    if (strncmp(arch.c_str(), "gfx950", 6) != 0)
        return fail(GBP_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only", h->device, arch.c_str());
    HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    h->flags = d->flags;


    Params &p = h->p;
    p.F = d->n_factors; p.L = d->n_lmks; p.C = C; p.T = 0;
    p.K = Intrinsics{d->K[0], d->K[1], d->K[2], d->K[3]};
    p.sigma2 = d->gauss_noise_std * d->gauss_noise_std;
    p.nstds = d->nstds; p.beta = d->beta; p.eta_damping = d->eta_damping;
    p.num_undamped = d->num_undamped_iters; p.min_linear = d->min_linear_iters; p.loss = d->loss;
    p.robustify = 0; p.local_relin = 1;
    p.crow = d->num_undamped_iters == 0 ? CSTAGE_ROW : CSTAGE_PLAIN;      // (xtra rows carry the dense remainder too)
    if (d->num_undamped_iters > ITERS_MAX || d->min_linear_iters > ITERS_MAX)
        return fail(GBP_EINVAL, "num_undamped_iters / min_linear_iters above %d are not supported (iters_since_relin saturates there)", ITERS_MAX);
    if (C >= (1 << (32 - META_LMK_BITS))) return fail(GBP_EINVAL, "more than %d cameras are not supported", (1 << (32 - META_LMK_BITS)) - 1);

    h->n_cus = n_cus;
    std::vector<void *> scratch;                             // device buffers only the build needs
    const int rc = build_graph(h, d, scratch, n_cus);
    for (void *q : scratch) (void)hipFreeAsync(q, h->stream);
    (void)hipStreamSynchronize(h->stream);
    return rc;
}

int gbp_ba_create(gbp_ba_t **out, const gbp_ba_desc_t *d)
{
    if (!out || !d) return fail(GBP_EINVAL, "null argument");
    *out = nullptr;
    if (d->n_cams < 0 || d->n_lmks < 0 || d->n_factors < 0) return fail(GBP_EINVAL, "negative size");
    if (d->n_factors > 0 && (!d->meas || !d->cam_idx || !d->lmk_idx)) return fail(GBP_EINVAL, "null observation arrays");
    if ((d->n_cams > 0 && !d->cam_means) || (d->n_lmks > 0 && !d->lmk_means)) return fail(GBP_EINVAL, "null initial means");
    if (d->loss < GBP_LOSS_NONE || d->loss > GBP_LOSS_CONSTANT) return fail(GBP_EINVAL, "unknown loss %d", d->loss);
    if (!(d->gauss_noise_std > 0)) return fail(GBP_EINVAL, "gauss_noise_std must be positive");
    gbp_ba *h = new (std::nothrow) gbp_ba;
    if (!h) return fail(GBP_ENOMEM, "out of host memory");
    int rc;
    try {
        rc = create_impl(h, d);
    } catch (const std::bad_alloc &) {
        rc = fail(GBP_ENOMEM, "out of host memory");
    }
    if (rc != GBP_OK) { std::string keep = g_err; gbp_ba_destroy(h); g_err = keep; return rc; }
    *out = h;
    return GBP_OK;
}

int gbp_ba_set_stream(gbp_ba_t *h, void *hip_stream)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return GBP_OK;
}

// A finish wave of the peer-store exchange that gave up waiting for a peer's partial sums leaves a mark; everything that hands results
// to the caller (sync, beliefs, means, are / energy, checkpoints) looks at it first, so a timed-out sweep cannot pass for a result.
// The mark stays until gbp_ba_sync has reported it.
}  // extern "C"
int gbp::peer_check(gbp_ba *h, bool clear)
{
    if (!h->peer.connected || !h->peer.d_ctl) return GBP_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    int err = 0;
    HIPCHK(hipMemcpy(&err, h->peer.d_ctl + 1, sizeof(int), hipMemcpyDeviceToHost));
    if (!err) return GBP_OK;
    if (clear) HIPCHK(hipMemset(h->peer.d_ctl + 1, 0, sizeof(int)));
    return fail(GBP_ESTATE, "peer-store exchange timed out: a rank's camera partial sums did not arrive (the camera beliefs since then are invalid)");
}

extern "C" {
int gbp_ba_sync(gbp_ba_t *h)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    return peer_check(h, true);
}

// ------------------------------------------------------------------------------- priors ---

// max over the adjacent factors of every variable of max(Lambda_f) (gbp_ba.py:27-31) into d_varmax = cameras | landmarks
static int variable_lambda_max(gbp_ba *h)
{
    const Params &p = h->p;
    const size_t S = n_slots(h);
    CHK(ensure_tmp(h, sizeof(double) * S));
    if (p.T) hipLaunchKernelGGL(k_factor_lambda_max, dim3(grid_for(S)), dim3(BLOCK), 0, h->stream, p, h->d_tmp);
    if (p.C) hipLaunchKernelGGL(k_cam_max, dim3(p.C), dim3(BLOCK), 0, h->stream, p, h->d_tmp, h->d_varmax);
    if (p.L) hipLaunchKernelGGL(k_lmk_max, dim3(grid_for((size_t)p.L)), dim3(BLOCK), 0, h->stream, p, h->d_tmp, h->d_varmax + p.C);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

int gbp_ba_factor_lambda_max(gbp_ba_t *h, double *cam_max, double *lmk_max)
{
    ENTER(h);
    const Params &p = h->p;
    CHK(variable_lambda_max(h));
    if (cam_max && p.C) HIPCHK(hipMemcpyAsync(cam_max, h->d_varmax, sizeof(double) * (size_t)p.C, hipMemcpyDeviceToHost, h->stream));
    if (lmk_max && p.L) HIPCHK(hipMemcpyAsync(lmk_max, h->d_varmax + p.C, sizeof(double) * (size_t)p.L, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return GBP_OK;
}

int gbp_ba_set_prior_scalars(gbp_ba_t *h, const double *cam_lambda, const double *lmk_lambda)
{
    ENTER(h);
    h->resid_ok = false;
    const Params &p = h->p;
    if (!cam_lambda || !lmk_lambda) return fail(GBP_EINVAL, "null argument");
    if (p.C) HIPCHK(hipMemcpyAsync(h->d_varmax, cam_lambda, sizeof(double) * (size_t)p.C, hipMemcpyHostToDevice, h->stream));
    if (p.L) HIPCHK(hipMemcpyAsync(h->d_varmax + p.C, lmk_lambda, sizeof(double) * (size_t)p.L, hipMemcpyHostToDevice, h->stream));
    if (p.C + p.L) hipLaunchKernelGGL(k_prior_scalars, dim3(grid_for((size_t)p.C + p.L)), dim3(BLOCK), 0, h->stream, p, h->d_varmax, h->d_varmax + p.C, 1.0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));               // the arrays are the caller's
    return GBP_OK;
}

int gbp_ba_generate_priors(gbp_ba_t *h, double weaker_factor)
{
    ENTER(h);
    h->resid_ok = false;
    if (!(weaker_factor != 0.0)) return fail(GBP_EINVAL, "weaker_factor must be non-zero");
    const Params &p = h->p;
    CHK(variable_lambda_max(h));                           // nothing F-sized leaves the device
    if (p.C + p.L) hipLaunchKernelGGL(k_prior_scalars, dim3(grid_for((size_t)p.C + p.L)), dim3(BLOCK), 0, h->stream, p, h->d_varmax,
                                      h->d_varmax + p.C, weaker_factor * weaker_factor);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

int gbp_ba_set_priors(gbp_ba_t *h, const double *cam_eta, const double *cam_lam, const double *lmk_eta, const double *lmk_lam)
{
    ENTER(h);
    h->resid_ok = false;
    const Params &p = h->p;
    if (!cam_eta || !cam_lam || !lmk_eta || !lmk_lam) return fail(GBP_EINVAL, "null argument");
    std::vector<double> cp((size_t)std::max(p.C, 1) * 27, 0.0), lp((size_t)std::max(p.L, 1) * 9, 0.0);
    for (int c = 0; c < p.C; ++c) {
        for (int k = 0; k < 6; ++k) cp[(size_t)c * 27 + k] = cam_eta[(size_t)c * 6 + k];
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j)
                cp[(size_t)c * 27 + 6 + Sym<6>::at(i, j)] = 0.5 * (cam_lam[(size_t)c * 36 + i * 6 + j] + cam_lam[(size_t)c * 36 + j * 6 + i]);
    }
    for (int l = 0; l < p.L; ++l) {
        for (int k = 0; k < 3; ++k) lp[(size_t)l * 9 + k] = lmk_eta[(size_t)l * 3 + k];
        for (int i = 0; i < 3; ++i)
            for (int j = i; j < 3; ++j)
                lp[(size_t)l * 9 + 3 + Sym<3>::at(i, j)] = 0.5 * (lmk_lam[(size_t)l * 9 + i * 3 + j] + lmk_lam[(size_t)l * 9 + j * 3 + i]);
    }
    CHK(upload(h, p.cprior, cp));
    CHK(ensure_tmp(h, sizeof(double) * lp.size()));
    CHK(upload(h, h->d_tmp, lp));
    if (p.L) hipLaunchKernelGGL(k_scatter_lmk_priors, dim3(grid_for((size_t)p.L * 9)), dim3(BLOCK), 0, h->stream, p, h->d_tmp);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

int gbp_ba_weaken_priors(gbp_ba_t *h, double factor)
{
    ENTER(h);
    h->resid_ok = false;
    const Params &p = h->p;
    const size_t n = (size_t)p.C * 27 + (size_t)p.L * 9;
    if (n) hipLaunchKernelGGL(k_weaken_priors, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, p, factor);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

// ------------------------------------------------------------------------------ BAL files ---
// utils/read_balfile.py:4-37 (called from create_ba_graph gbp_ba.py:108-109); host only.

int gbp_bal_header(const char *path, int32_t *n_cams, int32_t *n_lmks, int32_t *n_obs)
{
    if (!path || !n_cams || !n_lmks || !n_obs) return fail(GBP_EINVAL, "NULL argument");
    BalText t;
    std::string err;
    long C, L, F;
    if (t.open(path, err) || bal_header(t, C, L, F, err)) return fail(GBP_EINVAL, "%s: %s", path, err.c_str());
    if (C > INT32_MAX || L > INT32_MAX || F > INT32_MAX) return fail(GBP_EINVAL, "%s: sizes exceed int32", path);
    *n_cams = (int32_t)C; *n_lmks = (int32_t)L; *n_obs = (int32_t)F;
    return GBP_OK;
}

int gbp_bal_read(const char *path, int32_t n_cams, int32_t n_lmks, int32_t n_obs, double *K4, double *cam_means,
                 double *lmk_means, double *meas, int32_t *cam_idx, int32_t *lmk_idx)
{
    if (!path || !K4 || !cam_means || !lmk_means || !meas || !cam_idx || !lmk_idx) return fail(GBP_EINVAL, "NULL argument");
    BalText t;
    std::string err;
    long C, L, F;
    if (t.open(path, err) || bal_header(t, C, L, F, err)) return fail(GBP_EINVAL, "%s: %s", path, err.c_str());
    if (C != n_cams || L != n_lmks || F != n_obs) return fail(GBP_EINVAL, "%s: sizes differ from gbp_bal_header's", path);
    if (bal_body(t, C, L, F, K4, cam_means, lmk_means, meas, cam_idx, lmk_idx, err)) return fail(GBP_EINVAL, "%s: %s", path, err.c_str());
    return GBP_OK;
}

int gbp_ba_check_layout(gbp_ba_t *h, int32_t *bad_slots)
{
    ENTER(h);
    if (!bad_slots) return fail(GBP_EINVAL, "null argument");
    *bad_slots = 0;
    if (!h->p.T) return GBP_OK;
    HIPCHK(hipMemsetAsync(h->d_count, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_check_layout, dim3(grid_for(n_slots(h))), dim3(BLOCK), 0, h->stream, h->p, h->d_ref_cam, h->d_ref_lmk, h->d_count);
    HIPCHK(hipGetLastError());
    int v = 0;
    HIPCHK(hipMemcpyAsync(&v, h->d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *bad_slots = v;
    return GBP_OK;
}

int gbp_ba_fused_max_cams(void)
{
    return fused_max_cams();
}

int gbp_ba_plan_info(gbp_ba_t *h, int32_t *out, int32_t n)
{
    if (!h || (n > 0 && !out)) return fail(GBP_EINVAL, "null argument");
    const bool pinned = h->fused.enabled && h->fused.args.pin != 0x7fffffff;
    const int32_t v[GBP_PLAN_INFO_FIELDS] = {
        h->fused.enabled ? 1 : 0, h->staged_auto ? 1 : 0, h->fused.enabled ? h->fused.single : 0, h->fused.single_probe,
        pinned ? h->fused.args.pin : -1, h->fused.enabled ? h->fused.n_blocks : std::max(1, std::min(h->p.T, h->n_cus)), h->p.T,
        h->pack_mode, h->fused.enabled && h->fused.windowed ? h->fused.max_window : 0,
        h->fused.enabled ? (int32_t)std::min<long long>(h->fused.windowed ? h->fused.table_rows : (long long)h->fused.n_blocks * h->p.C, INT32_MAX) : 0,
        h->fused.enabled && h->fused.windowed ? h->fused.rows_wave : 0};
    for (int i = 0; i < n && i < GBP_PLAN_INFO_FIELDS; ++i) out[i] = v[i];
    return GBP_OK;
}

int gbp_ba_info(gbp_ba_t *h, int32_t *fused_path, int32_t *n_tiles, int32_t *n_blocks)
{
    if (!h) return fail(GBP_EINVAL, "null handle");
    if (fused_path) *fused_path = h->fused.enabled ? h->fused.n_groups : 0;
    if (n_tiles) *n_tiles = h->p.T;
    if (n_blocks) *n_blocks = h->fused.n_blocks;
    return GBP_OK;
}


}  // extern "C"

// digest of the layout: a state blob restores only into a handle of the same graph
int gbp::graph_hash(gbp_ba *h, uint64_t *out)
{
    if (!h->hash_ok) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(h->d_count);      // 8 bytes
        HIPCHK(hipMemsetAsync(d, 0, sizeof(unsigned long long), h->stream));
        if (h->p.F) hipLaunchKernelGGL(k_graph_hash, dim3(grid_for((size_t)h->p.F)), dim3(BLOCK), 0, h->stream, h->p.cadj, h->d_ref_cam,
                                       h->d_ref_lmk, h->p.F, d);
        HIPCHK(hipGetLastError());
        unsigned long long v = 0;
        HIPCHK(hipMemcpyAsync(&v, d, sizeof v, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->hash = v; h->hash_ok = true;
    }
    *out = h->hash;
    return GBP_OK;
}
