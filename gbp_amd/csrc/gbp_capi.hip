// gbp_capi.hip -- host side of libgbp_hip.so: graph lay-out, launches and the C ABI of include/gbp_ba.h.
//
// No CPU compute path lives here: every sweep, belief update and diagnostic is a HIP kernel, and so is the graph build
// (gbp_build.hpp: ordering, tile packing, slot assignment, initial linearisation points, prior maxima).  Host code only
// (a) sizes the allocations from the tile count the device reports, (b) packs/unpacks symmetric matrices for the views,
// (c) owns handles, streams and the optional RCCL communicator.
#include "../../include/gbp_ba.h"
#include "gbp_kernels.hpp"
#include "gbp_fused.hpp"
#include "gbp_build.hpp"
#include "gbp_balio.hpp"

#include <rccl/rccl.h>      // types only: the library is dlopen()ed when a communicator is asked for (no link-time dependency)
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace gbp;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace gbp {
// shared with gbp_lin_capi.hip
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

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(e__ == hipErrorOutOfMemory ? GBP_ENOMEM : GBP_EHIP, "%s failed: %s (%s:%d)",   \
                        #expr, hipGetErrorString(e__), __FILE__, __LINE__);                            \
    } while (0)

#define CHK(expr) do { int rc__ = (expr); if (rc__ != GBP_OK) return rc__; } while (0)


struct gbp_ba {
    Params p{};
    int device = 0;
    int flags = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // order maps: on the device (built there, gbp_build.hpp); the host keeps only what is L- or C-sized
    int *d_ref_cam = nullptr, *d_ref_lmk = nullptr;   // per reference factor (p.cadj = reference id -> slot, p.cpos = slot -> reference id)
    std::vector<int32_t> big_lmks;               // landmarks larger than a tile
    int *d_big = nullptr;                        // the same on the device (general sweep)
    bool hash_ok = false; uint64_t hash = 0;     // digest of the layout (state blobs)
    void *arena = nullptr; size_t arena_bytes = 0, arena_used = 0;
    std::vector<void *> snap; std::vector<size_t> snap_bytes; bool snap_has_beliefs = false; uint32_t snap_parity = 0; int snap_clk = 0;   // device-resident checkpoint (gbp_ba_snapshot_state)
    // device scratch
    double *d_partial = nullptr;                 // C*27 camera partial sums (single-GPU path)
    double *d_red = nullptr;                     // per-block residual partials
    double *d_tmp = nullptr; size_t tmp_bytes = 0;
    std::vector<void *> allocs;
    bool has_beliefs = false;
    int n_cus = 0;
    bool staged_auto = false;                    // the general sweep was picked by the sparseness rule (build_graph), not asked for
    bool staged_attr_set = false;                // k_sweep_staged's dynamic-LDS attribute has been set on this handle's device (staged_launch)
    bool pending_possible = false;               // a stage-wise relinearise / compute_factors has run since the messages were last computed
    // dense message remainder allocated on demand (a damped factor that moves its linearisation point: enable_remainder)
    double *xtra_buf = nullptr;                  // the allocation behind p.xtra when it was made after create
    bool lazy_xtra = false, fused_suspended = false;
    int cstage_cap = 0;                          // doubles per row the staging buffer was allocated for (0: not allocated yet)
    bool cstage_x0_ok = false;                   // the x0 halves of the staged rows are those of the factors' present linearisation points
                                                 // (only the staged sweep keeps them so: it then rewrites them for relinearising tiles alone)
    long lazy_since = 0;                         // sweeps run since the remainder was switched on (it is checked for all-zero every 16)
    bool resid_ok = false; double resid[2] = {0.0, 0.0};     // ARE / energy sums of the CURRENT state (ba.py asks for both every sweep)
    // streaming means export (viewer): device staging, two pinned host mirrors, a copy stream
    double *d_mu = nullptr, *h_mu[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_landed[2] = {nullptr, nullptr};
    long snap_count = 0;
    // fused path
    FusedPlan fused;
    // timing of the dominant kernel
    bool timing = false;
    int timing_every = 1, timing_tick = 0;       // events around every n-th launch of the dominant kernel (two event
                                                 // records per sweep cost ~6 us of a 125 us sweep)
    bool timing_now = false;
    std::vector<hipEvent_t> ev;                  // pairs
    size_t ev_used = 0;
    const char *dominant = "k_factor_tile";
    // device-clock stamps of instrumented sweeps: [CLK_RING][6] = {sweep start, end, reduce start, end, finish start, end}
    unsigned long long *d_clk = nullptr, *clk_cur = nullptr;
    int clk_used = 0, clk_rate_khz = 0;
    bool clk_calibrated = false; double clk_rate_khz_measured = 0.0;      // the counter's real rate (gbp_ba_set_kernel_timing)
    // per-sweep count of relinearising factors: ring of device counters, half of it cleared whenever the sweep index
    // enters it, so the last RELIN_RING/2 sweeps are always readable
    int *d_relin_ring = nullptr;
    long sweep_count = 0;                        // sweeps since create (index into the relin ring)
    uint32_t gen_parity = 0;                     // general sweep: direction of the walk (not part of the state: the sums do not depend on it)
    uint32_t walk_parity = 0;                    // part of the STATE: odd sweeps walk the tiles backwards, so a restored handle must
                                                 // resume with the parity it was saved with to continue bit-identically
    int *d_count = nullptr;                      // scratch counter of gbp_ba_count_relinearising / gbp_ba_check_layout
    double *d_varmax = nullptr;                  // C + L: per-variable max of Lambda_f, or the prior scalars on their way in
    // landmark-sharded sweep: the camera exchange (include/gbp_ba.h gbp_ba_set_exchange / gbp_ba_comm_init_rccl)
    gbp_exchange_fn xch_fn = nullptr;
    void *xch_ctx = nullptr;
    int xch_rank = 0, xch_ranks = 1, xch_flags = 0;
    double *d_send = nullptr, *d_recv = nullptr; // C*27 and n_ranks*C*27
    ncclComm_t comm = nullptr;
    // peer-store exchange (gbp_ba_peer_export / gbp_ba_peer_connect): this rank's mailbox and the peers' mapped ones
    struct Peer {
        void *mailbox = nullptr; bool finegrained = false;
        int n_ranks = 0, rank = 0; bool connected = false;
        void *base[MAX_PEERS] = {}; bool opened[MAX_PEERS] = {};
        unsigned long long seq = 0;
        int *d_ctl = nullptr;                    // {unused, err, selftest code, selftest rank}: a finish wave that gave up waiting sets err
        unsigned long long probe_seq = 0;        // self-tests run so far (every rank runs the same number)
        long long timeout_ticks = 0;
    } peer;
    hipStream_t side_stream = nullptr;           // beliefs of over-sized landmarks run beside the exchange
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};
constexpr int RELIN_RING = 1024;
constexpr int CLK_RING = 4096;

template <typename T>
static int dev_alloc(gbp_ba *h, T **out, size_t n, bool zero = true)
{
    void *ptr = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    const size_t off = (h->arena_used + 4095) & ~(size_t)4095;
    if (h->arena && off + bytes <= h->arena_bytes) {    // what a sweep streams lives in ONE allocation (see arena_reserve)
        ptr = static_cast<char *>(h->arena) + off;
        h->arena_used = off + bytes;
    } else {
        HIPCHK(hipMalloc(&ptr, bytes));
        h->allocs.push_back(ptr);
    }
    if (zero) HIPCHK(hipMemsetAsync(ptr, 0, bytes, h->stream));
    *out = static_cast<T *>(ptr);
    return GBP_OK;
}

// One allocation for everything a sweep streams (factor streams, landmark records, the workgroup tables, the small per-camera
// and control buffers).  Not a convenience: at the headline size the working set of a sweep (243 MB) is about the size of the
// 256 MiB Infinity Cache, and the SAME kernel on the SAME data ran 87 or 95-105 us per sweep depending on where a dozen separate
// hipMalloc blocks happened to land (one engine in four in the slow mode, tools/placement_probe.py); out of one block the slow
// mode becomes rare.  It also saves a dozen allocation calls (most of what is left of gbp_ba_create's time).
static int arena_reserve(gbp_ba *h, size_t bytes)
{
    if (h->arena || getenv("GBP_NO_ARENA")) return GBP_OK;
    HIPCHK(hipMalloc(&h->arena, bytes));
    h->allocs.push_back(h->arena);
    h->arena_bytes = bytes;
    h->arena_used = 0;
    return GBP_OK;
}

static void *arena_take(void *ctx, size_t bytes)          // FusedPlan's allocator hook
{
    gbp_ba *h = static_cast<gbp_ba *>(ctx);
    const size_t off = (h->arena_used + 4095) & ~(size_t)4095;
    if (!h->arena || off + bytes > h->arena_bytes) return nullptr;
    h->arena_used = off + bytes;
    return static_cast<char *>(h->arena) + off;
}

static int ensure_tmp(gbp_ba *h, size_t bytes)
{
    if (bytes <= h->tmp_bytes) return GBP_OK;
    if (h->d_tmp) { HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(hipFree(h->d_tmp)); h->d_tmp = nullptr; h->tmp_bytes = 0; }
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->d_tmp), bytes));
    h->tmp_bytes = bytes;
    return GBP_OK;
}

template <typename T>
static int upload(gbp_ba *h, T *dst, const std::vector<T> &src)
{
    if (src.empty()) return GBP_OK;
    HIPCHK(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));      // src is a temporary
    return GBP_OK;
}

template <typename T>
static int download(gbp_ba *h, std::vector<T> &dst, const T *src, size_t n)
{
    dst.resize(n);
    if (!n) return GBP_OK;
    HIPCHK(hipMemcpyAsync(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return GBP_OK;
}

static inline int grid_for(size_t n) { return (int)((n + BLOCK - 1) / BLOCK); }

// The relinearisation clock (gbp_kernels.hpp, state word): every call in which the reference's relinearise_factors() runs -- a sweep
// with local_relin, or the stage call itself -- advances it by one; a factor's iters_since_relin is the clock minus the value its
// state word holds.  Set before the launch: the kernels read the value AFTER the call's advance (Params::clk) and whether it advanced.
static inline void clock_tick(gbp_ba *h, bool advance)
{
    h->p.clk_inc = advance ? 1 : 0;
    if (advance) h->p.clk = (int)(((unsigned)h->p.clk + 1u) & CLK_MASK);
}

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

static int launch_cam_finish(gbp_ba *h, const double *gathered, int n_parts, size_t stride, const PeerWait *wait = nullptr)
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
    if (!h->big_lmks.empty() && !h->d_big) {
        CHK(dev_alloc(h, &h->d_big, h->big_lmks.size(), false));
        CHK(upload(h, h->d_big, h->big_lmks));
    }
    return GBP_OK;
}

static int launch_big_lmk_beliefs(gbp_ba *h, hipStream_t stream)
{
    const int *list = h->fused.enabled ? h->fused.d_big : h->d_big;
    const int n = (int)h->big_lmks.size();
    if (!n || !list) return GBP_OK;
    hipLaunchKernelGGL(k_lmk_belief_list, dim3((n + 63) / 64), dim3(64), 0, stream, h->p, list, n);
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

// defer_big: leave the beliefs of the over-sized landmarks (k_lmk_belief_list) to the caller, who runs them beside the
// camera exchange (launch_big_lmk_beliefs)
static int sweep_begin(gbp_ba *h, int with_messages, int robustify, int local_relin, double *partial, int finish = 0,
                       bool *finished = nullptr, bool defer_big = false, const PeerOut *peer = nullptr, const PeerWait *merged = nullptr)
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
        if (!defer_big) CHK(launch_big_lmk_beliefs(h, h->stream));
        if (h->p.C) {
            // One workgroup per camera.  Short runs (a camera with a few hundred factors: graphs with thousands of cameras) leave most
            // of a 256-thread block idle through its reduction and 6x6 solve: 128 threads do 1M factors x 2 000 / 3 000 cameras in
            // 125.6 / 127.4 us per sweep against 134.5 / 143.7, 3M factors x 13 682 cameras in 439 against 509; from ~700 factors per
            // camera on the two are equal, at 2 000 per camera 256 threads win (122.9 against 128.0).  The block size fixes the order of
            // the sums, so it depends on the graph's shape alone (GBP_CAM_BLOCK overrides, experiments).
            static const int forced = getenv("GBP_CAM_BLOCK") ? atoi(getenv("GBP_CAM_BLOCK")) : 0;
            // (and one wave per camera below 200 factors per camera: 1M factors x 20 000 cameras 176 against 243 us, 200k x 5 000 50.5 against 65.8)
            const int cam_block = forced ? forced : ((long long)h->p.F < 200LL * h->p.C ? 64 : (long long)h->p.F < 640LL * h->p.C ? 128 : BLOCK);
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

// ------------------------------------------------------------------------------- RCCL ------
// Resolved at run time: a process that never shards never maps librccl.  When PyTorch is in the process its bundled
// librccl.so is already mapped (and is the build that matches the HIP runtime torch brought along, see
// gbp_amd/_capi.py), so that one is taken; otherwise the system library.

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load(const char *path)
{
    static std::mutex load_mutex;                            // ranks as threads of one process may arrive together
    std::lock_guard<std::mutex> lock(load_mutex);
    if (g_rccl.lib) return GBP_OK;
    void *lib = nullptr;
    if (path && *path) lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !lib; ++pass)              // first whatever the process already holds, then a fresh load
        for (const char *nm : names) {
            lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (lib) break;
        }
    if (!lib) return fail(GBP_ESTATE, "librccl.so could not be loaded: %s", dlerror());
    Rccl r;
    r.lib = lib;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(lib, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(lib, "ncclCommCount"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString)
        return fail(GBP_ESTATE, "librccl.so lacks an expected entry point");
    g_rccl = r;
    return GBP_OK;
}

// gbp_exchange_fn over an RCCL communicator: one all-gather of C*27 doubles per rank, in stream order
int rccl_exchange(void *ctx, const double *send_dev, double *recv_dev, uint64_t count, void *stream)
{
    gbp_ba *h = static_cast<gbp_ba *>(ctx);
    const ncclResult_t rc = g_rccl.AllGather(send_dev, recv_dev, (size_t)count, ncclDouble, h->comm, static_cast<hipStream_t>(stream));
    if (rc != ncclSuccess) return fail(GBP_EHIP, "ncclAllGather failed: %s", g_rccl.GetErrorString(rc));
    return GBP_OK;
}
}  // namespace

// mailbox geometry: [2 halves][n_ranks][C] rows of PEER_ROW doubles (27 sums | tag), then [n_ranks] probe rows (gbp_ba_peer_selftest)
static inline size_t peer_block(const gbp_ba *h) { return (size_t)std::max(h->p.C, 1) * PEER_ROW; }
static inline size_t peer_bytes(const gbp_ba *h, int n) { return (2 * (size_t)n * peer_block(h) + (size_t)n * PEER_ROW) * sizeof(double); }   // + the self-test's probe rows
static inline double *peer_probe(const gbp_ba *h, void *base, int n, int src) { return static_cast<double *>(base) + 2 * (size_t)n * peer_block(h) + (size_t)src * PEER_ROW; }
static inline double *peer_data(const gbp_ba *h, void *base, int n, int half, int src)
{
    return static_cast<double *>(base) + ((size_t)half * n + src) * peer_block(h);
}

static void peer_release(gbp_ba *h)
{
    gbp_ba::Peer &pe = h->peer;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int r = 0; r < MAX_PEERS; ++r) {
        if (pe.opened[r] && pe.base[r]) (void)hipIpcCloseMemHandle(pe.base[r]);
        pe.opened[r] = false; pe.base[r] = nullptr;
    }
    pe.connected = false;
}

static void shard_comm_release(gbp_ba *h)
{
    if (h->comm && g_rccl.CommDestroy) { (void)hipStreamSynchronize(h->stream); (void)g_rccl.CommDestroy(h->comm); }
    h->comm = nullptr;
    if (h->xch_fn == rccl_exchange) { h->xch_fn = nullptr; h->xch_ctx = nullptr; h->xch_ranks = 1; h->xch_rank = 0; }
}

// ------------------------------------------------------------------------------- C ABI ----

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
    // 4. tiles: up to 64 slots / TILE_LMKS whole landmarks each; over-sized landmarks become chunk tiles (nl = 0).
    //    Next-fit packing as list ranking on the device (gbp_build.hpp); only the tile count and the (few) over-sized landmarks
    //    come back.
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
    int T = 0;
    HIPCHK(hipMemcpyAsync(&T, pos + L, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (L == 0) T = 0;
    if (T < 0 || (int64_t)T * WTILE > INT32_MAX) return fail(GBP_EINVAL, "the graph needs %d tiles: slot indices would not fit 32 bits", T);
    const size_t S = std::max<size_t>((size_t)T * WTILE, 1);
    p.T = T;
    int4 *d_tiles = nullptr;
    CHK(dev_alloc(h, &d_tiles, std::max<size_t>((size_t)T, 1)));
    if (L) hipLaunchKernelGGL(k_pack_emit, dim3(grid_for((size_t)L)), dim3(BLOCK), 0, h->stream, lptr, L, nxt, pos, d_tiles, d_lrow0, d_lrow1,
                              d_big_list, d_cnt);
    HIPCHK(hipGetLastError());
    int n_big = 0;
    HIPCHK(hipMemcpyAsync(&n_big, d_cnt, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (n_big) {
        CHK(download(h, h->big_lmks, d_big_list, (size_t)n_big));
        std::sort(h->big_lmks.begin(), h->big_lmks.end());                        // (the kernel appends them in any order)
    }

    clk.mark("tile packing");
    // 5. per-slot data
    bool general_sweep = false;
    {
        const int n_wg = std::max(1, std::min(T, n_cus));
        const int cgmax = fused_max_cams();
        // Few factors per camera: the fused sweep writes (and its reduce reads back) one 224-byte table row per camera and WORKGROUP
        // whatever the graph's size, the staged form one 128-byte row per FACTOR.  Below ~0.75 factors per (workgroup, camera) the
        // staged sweep is the faster one -- 13k / 30k / 60k / 90k factors x 500 cameras: 18.6 / 20.5 / 24.2 / 29.1 against 26.3 /
        // 28.6 / 28.9 / 30.2 us per sweep, 125k: 35.8 against 32.8 (profiles/r04_shards.json) -- e.g. a rank's share at 16 ranks and
        // beyond.  GBP_STAGED_BELOW overrides the 0.75 (0: never).
        double staged_below = 0.75;
        if (const char *e = getenv("GBP_STAGED_BELOW")) staged_below = atof(e);
        const bool sparse = (double)F < staged_below * (double)n_wg * (double)C;
        h->staged_auto = sparse && !(h->flags & (GBP_FLAG_FORCE_FUSED | GBP_FLAG_NO_FUSED));
        if (h->staged_auto) h->flags |= GBP_FLAG_NO_FUSED;
        general_sweep = (h->flags & GBP_FLAG_NO_FUSED) || p.num_undamped == 0 || C > cgmax;   // its staging buffer is streamed every sweep too
        const size_t need = (general_sweep ? std::max<size_t>(Fz, 1) * p.crow * sizeof(double) + (64 << 8) : 0) + S * (LIN_ROWS + MSG_ROWS + (p.num_undamped == 0 ? XTRA_ROW : 0) + (p.loss != 0 ? 1 : 0)) * sizeof(double) + S * sizeof(int)
                          + (size_t)std::max(L, 1) * LREC * sizeof(double) + (size_t)n_wg * std::max(C, 1) * TROW * sizeof(double)
                          + (size_t)std::max(C, 1) * (CAMREC + CBEL + 27 + 27 + 1) * sizeof(double) + (size_t)std::max(L, 1) * sizeof(double)
                          + 2 * (size_t)grid_for(S) * sizeof(double) + (size_t)RELIN_RING * RELIN_LANES * sizeof(int)
                          + (size_t)(n_wg + 1 + h->big_lmks.size()) * sizeof(int) + (64 << 12);
        CHK(arena_reserve(h, need));
    }
    if (general_sweep && F > 0) { CHK(dev_alloc(h, &p.cstage, Fz * p.crow)); h->cstage_cap = p.crow; }   // out of the same arena (else: on first use, ensure_staging)
    CHK(dev_alloc(h, &p.lin, S * LIN_ROWS)); CHK(dev_alloc(h, &p.msg, S * MSG_ROWS));
    if (p.num_undamped == 0) CHK(dev_alloc(h, &p.xtra, S * XTRA_ROW));      // damped in the relinearising sweep: gbp_math.hpp header
    if (p.loss != 0) CHK(dev_alloc(h, &p.avar, S));         // adaptive variances: robust losses only
    CHK(dev_alloc(h, &cpos, S));
    CHK(dev_alloc(h, &p.lrec, (size_t)std::max(L, 1) * LREC));
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
        int rc = fused_plan(h->fused, p, h->big_lmks, h->stream, n_cus);
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

#define ENTER(h)                                                  \
    if (!(h)) return fail(GBP_EINVAL, "null handle");             \
    HIPCHK(hipSetDevice((h)->device))

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
static int peer_check(gbp_ba *h, bool clear)
{
    if (!h->peer.connected || !h->peer.d_ctl) return GBP_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    int err = 0;
    HIPCHK(hipMemcpy(&err, h->peer.d_ctl + 1, sizeof(int), hipMemcpyDeviceToHost));
    if (!err) return GBP_OK;
    if (clear) HIPCHK(hipMemset(h->peer.d_ctl + 1, 0, sizeof(int)));
    return fail(GBP_ESTATE, "peer-store exchange timed out: a rank's camera partial sums did not arrive (the camera beliefs since then are invalid)");
}

int gbp_ba_sync(gbp_ba_t *h)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    return peer_check(h, true);
}

// ------------------------------------------------------------------------------- priors ---

static inline size_t n_slots(const gbp_ba *h) { return std::max<size_t>((size_t)h->p.T * WTILE, 1); }

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

// -------------------------------------------------------------------------------- sweep ---

// ---- the dense message remainder on demand -------------------------------------------------------------------------------
// A message is stored as coefficients in the rows of its factor's Jacobian (gbp_math.hpp).  The one thing that does not fit is a
// factor that is DAMPED in the message computation that moves its linearisation point: d * (old eta) lies in the span of the OLD
// Jacobian.  The reference allows it at any time (compute_all_factors with damping on, gbp.py:60-62; relinearise_factors followed by
// compute_all_messages(local_relin=False), gbp.py:46-54); graphs created with num_undamped_iters = 0 carry the out-of-span part
// from the start (Params::xtra, 9 doubles per factor), every other graph gets it HERE, the first time such a call sequence
// shows up, and runs the general sweep (the kernels with the XTRA template flag) until every remainder has decayed to exactly
// zero again -- which the next undamped message of a factor does (x' = d x), i.e. after the next relinearisation wave.
static inline size_t n_slots(const gbp_ba *h);
static int enable_remainder(gbp_ba *h)
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
static int remainder_guard(gbp_ba *h, int local_relin, int no_test)
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
static int remainder_release(gbp_ba *h)
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

static int shard_buffers(gbp_ba *h, int n_ranks)
{
    const size_t n = (size_t)std::max(h->p.C, 1) * 27;
    if (h->d_send) { HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(hipFree(h->d_send)); HIPCHK(hipFree(h->d_recv)); h->d_send = h->d_recv = nullptr; }
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->d_send), n * sizeof(double)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->d_recv), n * sizeof(double) * (size_t)n_ranks));
    return GBP_OK;
}

int gbp_ba_set_exchange(gbp_ba_t *h, gbp_exchange_fn fn, void *ctx, int32_t rank, int32_t n_ranks, int32_t flags)
{
    ENTER(h);
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(GBP_EINVAL, "rank %d of %d", rank, n_ranks);
    if (!fn && n_ranks > 1) return fail(GBP_EINVAL, "an exchange function is needed for more than one rank");
    peer_release(h);                                         // (a connected peer-store exchange would keep routing the sweeps)
    shard_comm_release(h);
    CHK(shard_buffers(h, n_ranks));
    h->xch_fn = fn; h->xch_ctx = ctx; h->xch_rank = rank; h->xch_ranks = n_ranks; h->xch_flags = flags;
    return GBP_OK;
}

int gbp_ba_comm_unique_id(void *id128, const char *rccl_path)
{
    if (!id128) return fail(GBP_EINVAL, "null argument");
    CHK(rccl_load(rccl_path));
    ncclUniqueId id;
    const ncclResult_t rc = g_rccl.GetUniqueId(&id);
    if (rc != ncclSuccess) return fail(GBP_EHIP, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(rc));
    static_assert(sizeof(id) == GBP_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id128, &id, sizeof id);
    return GBP_OK;
}

int gbp_ba_comm_init_rccl(gbp_ba_t *h, const void *id128, int32_t rank, int32_t n_ranks, int32_t flags, const char *rccl_path)
{
    ENTER(h);
    if (!id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(GBP_EINVAL, "bad communicator arguments (rank %d of %d)", rank, n_ranks);
    CHK(rccl_load(rccl_path));
    peer_release(h);
    shard_comm_release(h);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    const ncclResult_t rc = g_rccl.CommInitRank(&h->comm, n_ranks, id, rank);
    if (rc != ncclSuccess) { h->comm = nullptr; return fail(GBP_EHIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(rc)); }
    CHK(shard_buffers(h, n_ranks));
    h->xch_fn = rccl_exchange; h->xch_ctx = h; h->xch_rank = rank; h->xch_ranks = n_ranks; h->xch_flags = flags;
    return GBP_OK;
}

int gbp_ba_comm_destroy(gbp_ba_t *h)
{
    ENTER(h);
    shard_comm_release(h);
    peer_release(h);
    h->xch_fn = nullptr; h->xch_ctx = nullptr; h->xch_ranks = 1; h->xch_rank = 0;
    if (h->d_send) { HIPCHK(hipStreamSynchronize(h->stream)); (void)hipFree(h->d_send); (void)hipFree(h->d_recv); h->d_send = h->d_recv = nullptr; }
    return GBP_OK;                                           // (gbp_ba_iterate_sharded now reports "no exchange set" instead of running on stale buffers)
}

int gbp_ba_peer_export(gbp_ba_t *h, int32_t n_ranks, void *handle64, int32_t flags)
{
    ENTER(h);
    if (!handle64 || n_ranks < 1 || n_ranks > MAX_PEERS) return fail(GBP_EINVAL, "peer exchange: 1..%d ranks", MAX_PEERS);
    gbp_ba::Peer &pe = h->peer;
    peer_release(h);
    if (pe.mailbox) { HIPCHK(hipFree(pe.mailbox)); pe.mailbox = nullptr; }
    const size_t bytes = peer_bytes(h, n_ranks);
    // fine-grained (uncached across devices) when the runtime grants it: peers store into it over xGMI while this rank polls it
    // Fine-grained (uncached across devices): peers store into it over xGMI while this rank polls it, and the protocol has no fences --
    // on coarse-grained pages a polling load may keep hitting a stale L2 line.  No silent fallback: GBP_PEER_COARSE=1 is a debug switch
    // for ranks that share ONE device.
    if (getenv("GBP_PEER_COARSE")) {
        pe.finegrained = false;
        HIPCHK(hipMalloc(&pe.mailbox, bytes));
    } else {
        const hipError_t fe = hipExtMallocWithFlags(&pe.mailbox, bytes, hipDeviceMallocFinegrained);
        if (fe != hipSuccess) {
            (void)hipGetLastError();
            pe.mailbox = nullptr;
            return fail(GBP_EHIP, "peer exchange: fine-grained device memory for the mailbox is not available (%s); use the RCCL exchange", hipGetErrorString(fe));
        }
        pe.finegrained = true;
    }
    HIPCHK(hipMemsetAsync(pe.mailbox, 0, bytes, h->stream));
    if (!pe.d_ctl) HIPCHK(hipMalloc(reinterpret_cast<void **>(&pe.d_ctl), 4 * sizeof(int)));
    HIPCHK(hipMemsetAsync(pe.d_ctl, 0, 4 * sizeof(int), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    pe.n_ranks = n_ranks; pe.seq = 0; pe.probe_seq = 0;
    std::memset(handle64, 0, GBP_PEER_HANDLE_BYTES);
    if (flags & GBP_PEER_SAME_PROCESS) {
        std::memcpy(handle64, &pe.mailbox, sizeof(void *));
    } else {
        hipIpcMemHandle_t ipc;
        static_assert(sizeof(ipc) <= GBP_PEER_HANDLE_BYTES, "hipIpcMemHandle_t size");
        HIPCHK(hipIpcGetMemHandle(&ipc, pe.mailbox));
        std::memcpy(handle64, &ipc, sizeof ipc);
    }
    return GBP_OK;
}

int gbp_ba_peer_connect(gbp_ba_t *h, int32_t rank, int32_t n_ranks, const void *handles, int32_t flags)
{
    ENTER(h);
    gbp_ba::Peer &pe = h->peer;
    if (!handles || !pe.mailbox || n_ranks != pe.n_ranks || rank < 0 || rank >= n_ranks)
        return fail(GBP_EINVAL, "peer exchange: connect needs the %d handles of gbp_ba_peer_export (rank %d of %d)", pe.n_ranks, rank, n_ranks);
    peer_release(h);
    shard_comm_release(h);
    const char *hs = static_cast<const char *>(handles);
    for (int r = 0; r < n_ranks; ++r) {
        if (r == rank) { pe.base[r] = pe.mailbox; continue; }
        if (flags & GBP_PEER_SAME_PROCESS) {
            std::memcpy(&pe.base[r], hs + (size_t)r * GBP_PEER_HANDLE_BYTES, sizeof(void *));
        } else {
            hipIpcMemHandle_t ipc;
            std::memcpy(&ipc, hs + (size_t)r * GBP_PEER_HANDLE_BYTES, sizeof ipc);
            HIPCHK(hipIpcOpenMemHandle(&pe.base[r], ipc, hipIpcMemLazyEnablePeerAccess));
            pe.opened[r] = true;
        }
        if (!pe.base[r]) return fail(GBP_EINVAL, "peer exchange: rank %d's mailbox handle is empty", r);
    }
    CHK(shard_buffers(h, 1));                                // d_send: the partial sums of the general sweep on their way to the mailboxes
    pe.rank = rank;
    double ms = 20000.0;                                     // how long a finish kernel waits for a peer before it gives up
    if (const char *e = getenv("GBP_PEER_TIMEOUT_MS")) ms = std::max(1.0, atof(e));
    int clk_khz = 0;
    if (hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || clk_khz <= 0) clk_khz = 100000;
    pe.timeout_ticks = (long long)(ms * (double)clk_khz);    // wall_clock64 ticks (100 MHz on MI355X)
    if (!(flags & GBP_PEER_RENDEZVOUS)) { h->xch_fn = nullptr; h->xch_ctx = nullptr; }
    h->xch_rank = rank; h->xch_ranks = n_ranks;
    pe.connected = true;
    return GBP_OK;
}

// Every rank calls this after gbp_ba_peer_connect -- and after a side-channel barrier, so that every mailbox is mapped everywhere -- and
// before the first sharded call: k_peer_selftest (gbp_kernels.hpp) sends one tagged row to every rank and checks the rows of all ranks.
int gbp_ba_peer_selftest(gbp_ba_t *h, int32_t timeout_ms)
{
    ENTER(h);
    gbp_ba::Peer &pe = h->peer;
    if (!pe.connected) return fail(GBP_ESTATE, "peer exchange: self-test before gbp_ba_peer_connect");
    const int n = pe.n_ranks;
    PeerOut po{};
    po.n = n; po.seq = 0x9b50000000000000ull | ++pe.probe_seq;
    for (int r = 0; r < n; ++r) po.dst[r] = peer_probe(h, pe.base[r], n, pe.rank);
    int clk_khz = 0;
    if (hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || clk_khz <= 0) clk_khz = 100000;
    const long long ticks = (long long)((double)std::max(1, timeout_ms) * (double)clk_khz);
    HIPCHK(hipMemsetAsync(pe.d_ctl + 2, 0, 2 * sizeof(int), h->stream));
    hipLaunchKernelGGL(k_peer_selftest, dim3(1), dim3(64), 0, h->stream, po, peer_probe(h, pe.mailbox, n, 0), pe.rank, ticks, pe.d_ctl + 2);
    HIPCHK(hipGetLastError());
    int res[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(res, pe.d_ctl + 2, sizeof res, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (res[0] & 2) return fail(GBP_ESTATE, "peer exchange self-test: the probe row of rank %d arrived in rank %d's mailbox with wrong contents", res[1], pe.rank);
    if (res[0] & 1) return fail(GBP_ESTATE, "peer exchange self-test: the probe row of rank %d did not reach rank %d within %d ms", res[1], pe.rank, timeout_ms);
    return GBP_OK;
}

// one sharded sweep (or belief update) on the handle's stream: local kernels -> camera partial sums -> exchange -> rank-ordered
// sum + prior + 6x6 solve.  With one rank and no GBP_XCH_ALWAYS nothing is exchanged and the camera beliefs are finished by
// the reduce launch itself, exactly like gbp_ba_iterate.
// one sharded sweep under the peer-store exchange: no collective, no host synchronisation -- the wave that finishes a camera's partial
// sums stores the row into every rank's mailbox and raises its tag; whoever finishes the camera waits for the n_ranks tags of its row
static int sharded_step_peer(gbp_ba *h, int with_messages, int robustify, int local_relin)
{
    gbp_ba::Peer &pe = h->peer;
    const int n = pe.n_ranks, half = (int)(++pe.seq & 1ull);
    PeerOut po{};
    po.n = n; po.seq = pe.seq;
    for (int r = 0; r < n; ++r) po.dst[r] = peer_data(h, pe.base[r], n, half, pe.rank);
    PeerWait w{peer_data(h, pe.mailbox, n, half, 0), pe.seq, pe.timeout_ticks, pe.d_ctl + 1, nullptr};
    const bool big = with_messages && !h->big_lmks.empty();
    // Without a rendezvous hook everything behind the fused sweep is ONE launch (k_cam_reduce_xchg); logical ranks on one device
    // (the hook is set) keep reduce / push and finish apart, with the hook between them, so that they never spin on each other.
    const bool merged = !h->xch_fn && with_messages && h->fused.enabled && !getenv("GBP_PEER_SPLIT");
    bool finished = false;
    CHK(sweep_begin(h, with_messages, robustify, local_relin, h->d_send, 0, &finished, big, &po, merged ? &w : nullptr));
    if (big) {
        if (!h->side_stream) {
            HIPCHK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
        CHK(launch_big_lmk_beliefs(h, h->side_stream));
        HIPCHK(hipEventRecord(h->ev_join, h->side_stream));
    }
    if (!finished) {
        if (h->xch_fn) {                                     // rendezvous hook (logical ranks on ONE device: tests)
            int rc = h->xch_fn(h->xch_ctx, nullptr, nullptr, 0, h->stream);
            if (rc != GBP_OK) return rc < 0 ? rc : fail(GBP_EHIP, "the rendezvous function returned %d", rc);
        }
        CHK(launch_cam_finish(h, nullptr, n, 0, &w));
    }
    if (big) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    return GBP_OK;
}

static int sharded_step(gbp_ba *h, int with_messages, int robustify, int local_relin)
{
    if (h->peer.connected) return sharded_step_peer(h, with_messages, robustify, local_relin);
    const bool exchange = h->xch_ranks > 1 || ((h->xch_flags & GBP_XCH_ALWAYS) && h->xch_fn);
    if (!exchange) {
        bool finished = false;
        CHK(sweep_begin(h, with_messages, robustify, local_relin, h->d_partial, 1, &finished));
        if (!finished) CHK(launch_cam_finish(h, h->d_partial, 1, 0));
        return GBP_OK;
    }
    const bool big = with_messages && !h->big_lmks.empty();     // their beliefs need nothing from the exchange: side stream
    CHK(sweep_begin(h, with_messages, robustify, local_relin, h->d_send, 0, nullptr, big));
    if (big) {
        if (!h->side_stream) {
            HIPCHK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
        CHK(launch_big_lmk_beliefs(h, h->side_stream));
        HIPCHK(hipEventRecord(h->ev_join, h->side_stream));
    }
    int rc = h->xch_fn(h->xch_ctx, h->d_send, h->d_recv, (uint64_t)h->p.C * 27, h->stream);
    if (rc != GBP_OK) return rc < 0 ? rc : fail(GBP_EHIP, "the exchange function returned %d", rc);
    CHK(launch_cam_finish(h, h->d_recv, h->xch_ranks, (size_t)h->p.C * 27));
    if (big) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    return GBP_OK;
}

int gbp_ba_iterate_sharded(gbp_ba_t *h, int32_t n_iters, int32_t robustify, int32_t local_relin)
{
    ENTER(h);
    h->resid_ok = false;
    if (n_iters < 0) return fail(GBP_EINVAL, "n_iters < 0");
    if (!h->d_send) return fail(GBP_ESTATE, "no exchange set (gbp_ba_comm_init_rccl / gbp_ba_set_exchange / gbp_ba_peer_connect)");
    for (int it = 0; it < n_iters; ++it) {
        CHK(remainder_guard(h, local_relin, 0));
        CHK(sharded_step(h, 1, robustify, local_relin));
        h->pending_possible = false;
        CHK(remainder_release(h));
    }
    h->has_beliefs = true;
    return GBP_OK;
}

int gbp_ba_update_beliefs_sharded(gbp_ba_t *h)
{
    ENTER(h);
    h->resid_ok = false;
    if (!h->d_send) return fail(GBP_ESTATE, "no exchange set (gbp_ba_comm_init_rccl / gbp_ba_set_exchange)");
    CHK(sharded_step(h, 0, 0, 0));
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

// --------------------------------------------------------------------------------- views ---

static void unpack6(const double *pk, double *dense) { for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) dense[i * 6 + j] = pk[Sym<6>::at(std::min(i, j), std::max(i, j))]; }
static void unpack3(const double *pk, double *dense) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dense[i * 3 + j] = pk[Sym<3>::at(std::min(i, j), std::max(i, j))]; }

// cameras: rows of `cam_stride` doubles with (eta 6 | Lambda 21) in front; landmarks: rows of `lmk_stride` doubles with (eta 3 | Lambda 6) at lmk_off
static int get_var_info(gbp_ba *h, const double *d_cam, int cam_stride, const double *d_lmk, int lmk_stride, int lmk_off,
                        double *cam_eta, double *cam_lam, double *lmk_eta, double *lmk_lam)
{
    const Params &p = h->p;
    if (cam_eta || cam_lam) {
        std::vector<double> cb;
        CHK(download(h, cb, d_cam, (size_t)std::max(p.C, 1) * cam_stride));
        for (int c = 0; c < p.C; ++c) {
            if (cam_eta) for (int k = 0; k < 6; ++k) cam_eta[(size_t)c * 6 + k] = cb[(size_t)c * cam_stride + k];
            if (cam_lam) unpack6(&cb[(size_t)c * cam_stride + 6], cam_lam + (size_t)c * 36);
        }
    }
    if (lmk_eta || lmk_lam) {
        std::vector<double> lr;
        CHK(download(h, lr, d_lmk, (size_t)std::max(p.L, 1) * lmk_stride));
        for (int l = 0; l < p.L; ++l) {
            if (lmk_eta) for (int k = 0; k < 3; ++k) lmk_eta[(size_t)l * 3 + k] = lr[(size_t)l * lmk_stride + lmk_off + k];
            if (lmk_lam) unpack3(&lr[(size_t)l * lmk_stride + lmk_off + 3], lmk_lam + (size_t)l * 9);
        }
    }
    return GBP_OK;
}

int gbp_ba_get_beliefs(gbp_ba_t *h, double *cam_eta, double *cam_lam, double *lmk_eta, double *lmk_lam)
{
    ENTER(h);
    CHK(peer_check(h, false));
    const Params &p = h->p;
    const double *d_lmk = nullptr;
    if (lmk_eta || lmk_lam) {
        // VariableNode.belief of the landmarks is a view formed from mean | covariance (k_lmk_belief_view); zeros before the first
        // update_all_beliefs, like the reference's freshly constructed nodes (gbp.py:164)
        CHK(ensure_tmp(h, sizeof(double) * 9 * (size_t)std::max(p.L, 1)));
        if (!h->has_beliefs) HIPCHK(hipMemsetAsync(h->d_tmp, 0, sizeof(double) * 9 * (size_t)std::max(p.L, 1), h->stream));
        else if (p.L) hipLaunchKernelGGL(k_lmk_belief_view, dim3(grid_for((size_t)p.L)), dim3(BLOCK), 0, h->stream, p, h->d_tmp);
        HIPCHK(hipGetLastError());
        d_lmk = h->d_tmp;
    }
    return get_var_info(h, p.cbelief, CBEL, d_lmk, 9, 0, cam_eta, cam_lam, lmk_eta, lmk_lam);
}

int gbp_ba_get_priors(gbp_ba_t *h, double *cam_eta, double *cam_lam, double *lmk_eta, double *lmk_lam)
{
    ENTER(h);
    return get_var_info(h, h->p.cprior, 27, h->p.lrec, LREC, LR_PRIOR, cam_eta, cam_lam, lmk_eta, lmk_lam);
}

int gbp_ba_get_means(gbp_ba_t *h, double *cam_mu, double *lmk_mu)
{
    ENTER(h);
    CHK(peer_check(h, false));
    const Params &p = h->p;
    if (cam_mu) {
        std::vector<double> cb;
        CHK(download(h, cb, p.cbel, (size_t)std::max(p.C, 1) * CAMREC));
        for (int c = 0; c < p.C; ++c) for (int k = 0; k < 6; ++k) cam_mu[(size_t)c * 6 + k] = cb[(size_t)c * CAMREC + CAM_MU + k];
    }
    if (lmk_mu) {
        std::vector<double> lr;
        CHK(download(h, lr, p.lrec, (size_t)std::max(p.L, 1) * LREC));
        for (int l = 0; l < p.L; ++l) for (int k = 0; k < 3; ++k) lmk_mu[(size_t)l * 3 + k] = lr[(size_t)l * LREC + LR_MU + k];
    }
    return GBP_OK;
}

int gbp_ba_get_covariances(gbp_ba_t *h, double *cam_sigma, double *lmk_sigma)
{
    ENTER(h);
    CHK(peer_check(h, false));
    const Params &p = h->p;
    if (!h->has_beliefs) return fail(GBP_ESTATE, "beliefs have not been computed yet (Sigma is zeros in the reference, gbp.py:166)");
    const size_t nc = (size_t)p.C * 21, nl = (size_t)p.L * 6;
    CHK(ensure_tmp(h, sizeof(double) * (nc + nl + 1)));
    if (p.C + p.L) hipLaunchKernelGGL(k_covariances, dim3(grid_for((size_t)p.C + p.L)), dim3(BLOCK), 0, h->stream, p, h->d_tmp, h->d_tmp + nc);
    HIPCHK(hipGetLastError());
    std::vector<double> s;
    CHK(download(h, s, h->d_tmp, nc + nl));
    if (cam_sigma) for (int c = 0; c < p.C; ++c) unpack6(&s[(size_t)c * 21], cam_sigma + (size_t)c * 36);
    if (lmk_sigma) for (int l = 0; l < p.L; ++l) unpack3(&s[nc + (size_t)l * 6], lmk_sigma + (size_t)l * 9);
    return GBP_OK;
}

static int check_range(gbp_ba *h, int32_t f0, int32_t n)
{
    if (f0 < 0 || n < 0 || (int64_t)f0 + n > h->p.F) return fail(GBP_EINVAL, "factor range [%d, %d) outside [0, %d)", f0, f0 + n, h->p.F);
    return GBP_OK;
}

int gbp_ba_get_messages(gbp_ba_t *h, int32_t f0, int32_t n, double *cam_eta, double *cam_lam, double *lmk_eta, double *lmk_lam)
{
    ENTER(h);
    CHK(check_range(h, f0, n));
    const Params &p = h->p;
    if (!n || !(cam_eta || cam_lam || lmk_eta || lmk_lam)) return GBP_OK;
    CHK(ensure_tmp(h, sizeof(double) * 36 * (size_t)n));
    hipLaunchKernelGGL(k_export_messages, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, p, p.cadj + f0, n, h->d_tmp);
    HIPCHK(hipGetLastError());
    std::vector<double> m;
    CHK(download(h, m, h->d_tmp, 36 * (size_t)n));
    for (int q = 0; q < n; ++q) {
        const double *o = &m[(size_t)q * 36];
        if (cam_eta) for (int k = 0; k < 6; ++k) cam_eta[(size_t)q * 6 + k] = o[k];
        if (cam_lam) unpack6(o + 6, cam_lam + (size_t)q * 36);
        if (lmk_eta) for (int k = 0; k < 3; ++k) lmk_eta[(size_t)q * 3 + k] = o[27 + k];
        if (lmk_lam) unpack3(o + 30, lmk_lam + (size_t)q * 9);
    }
    return GBP_OK;
}

int gbp_ba_get_factors(gbp_ba_t *h, int32_t f0, int32_t n, double *eta, double *lam, double *linpoint, int32_t *cam, int32_t *lmk, double *meas)
{
    ENTER(h);
    CHK(check_range(h, f0, n));
    const Params &p = h->p;
    if (cam && n) HIPCHK(hipMemcpyAsync(cam, h->d_ref_cam + f0, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    if (lmk && n) HIPCHK(hipMemcpyAsync(lmk, h->d_ref_lmk + f0, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    if ((cam || lmk) && n) HIPCHK(hipStreamSynchronize(h->stream));
    if ((linpoint || meas) && n) {                                  // gathered on the device: only the requested range moves
        CHK(ensure_tmp(h, sizeof(double) * 11 * (size_t)n));
        double *d_x0 = h->d_tmp, *d_z = h->d_tmp + 9 * (size_t)n;
        hipLaunchKernelGGL(k_export_lin, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, p, p.cadj + f0, n, linpoint ? d_x0 : nullptr,
                           meas ? d_z : nullptr);
        HIPCHK(hipGetLastError());
        if (linpoint) HIPCHK(hipMemcpyAsync(linpoint, d_x0, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        if (meas) HIPCHK(hipMemcpyAsync(meas, d_z, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if ((eta || lam) && n) {
        CHK(ensure_tmp(h, sizeof(double) * 90 * (size_t)n));
        hipLaunchKernelGGL(k_export_factors, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, p, p.cadj + f0, n, h->d_tmp, h->d_tmp + 9 * (size_t)n);
        HIPCHK(hipGetLastError());
        if (eta) HIPCHK(hipMemcpyAsync(eta, h->d_tmp, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        if (lam) HIPCHK(hipMemcpyAsync(lam, h->d_tmp + 9 * (size_t)n, sizeof(double) * 81 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return GBP_OK;
}

// the range [f0, f0+n) of the reference's factor order, gathered on the device (cadj = reference id -> slot)
static int relin_range(gbp_ba *h, int32_t f0, int32_t n, int32_t *iters, double *eta_damping, double *adaptive_var, uint8_t *robust_flag)
{
    const Params &p = h->p;
    if (!n) return GBP_OK;
    const size_t N = (size_t)n;
    CHK(ensure_tmp(h, N * (sizeof(double) + sizeof(int)) + N + 16));
    double *d_av = h->d_tmp;
    int *d_it = reinterpret_cast<int *>(h->d_tmp + N);
    unsigned char *d_fl = reinterpret_cast<unsigned char *>(d_it + N);
    hipLaunchKernelGGL(k_export_relin, dim3(grid_for(N)), dim3(BLOCK), 0, h->stream, p, p.cadj + f0, n, d_it, d_fl,
                       adaptive_var ? d_av : nullptr);
    HIPCHK(hipGetLastError());
    std::vector<uint8_t> fl(N);
    if (iters) HIPCHK(hipMemcpyAsync(iters, d_it, N * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (adaptive_var) HIPCHK(hipMemcpyAsync(adaptive_var, d_av, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(fl.data(), d_fl, N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t q = 0; q < N; ++q) {
        if (eta_damping) eta_damping[q] = (fl[q] & 1) ? p.eta_damping : 0.0;
        if (robust_flag) robust_flag[q] = (uint8_t)((fl[q] >> 1) & 1);
    }
    return GBP_OK;
}

int gbp_ba_get_relin_state(gbp_ba_t *h, int32_t *iters, double *eta_damping, double *adaptive_var, uint8_t *robust_flag)
{
    ENTER(h);
    return relin_range(h, 0, h->p.F, iters, eta_damping, adaptive_var, robust_flag);
}

int gbp_ba_get_relin_state_range(gbp_ba_t *h, int32_t f0, int32_t n, int32_t *iters, double *eta_damping, double *adaptive_var,
                                 uint8_t *robust_flag)
{
    ENTER(h);
    CHK(check_range(h, f0, n));
    return relin_range(h, f0, n, iters, eta_damping, adaptive_var, robust_flag);
}

int gbp_ba_set_iters_since_relin(gbp_ba_t *h, const int32_t *iters)
{
    ENTER(h);
    if (!iters) return fail(GBP_EINVAL, "null argument");
    const Params &p = h->p;
    for (int r = 0; r < p.F; ++r)
        if (iters[r] < 0 || iters[r] > ITERS_MAX) return fail(GBP_EINVAL, "iters_since_relin[%d] = %d outside [0, %d]", r, iters[r], ITERS_MAX);
    if (!p.F) return GBP_OK;
    CHK(ensure_tmp(h, sizeof(int) * (size_t)p.F));
    int *d_it = reinterpret_cast<int *>(h->d_tmp);
    HIPCHK(hipMemcpyAsync(d_it, iters, sizeof(int) * (size_t)p.F, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_import_iters, dim3(grid_for((size_t)p.F)), dim3(BLOCK), 0, h->stream, p, p.cadj, p.F, d_it);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));       // `iters` is the caller's
    return GBP_OK;
}

int gbp_ba_count_relinearising(gbp_ba_t *h, int64_t *count)
{
    ENTER(h);
    if (!count) return fail(GBP_EINVAL, "null argument");
    *count = 0;
    if (!h->p.T) return GBP_OK;
    HIPCHK(hipMemsetAsync(h->d_count, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_count_relin, dim3(grid_for(n_slots(h))), dim3(BLOCK), 0, h->stream, h->p, h->d_count);
    HIPCHK(hipGetLastError());
    int v = 0;
    HIPCHK(hipMemcpyAsync(&v, h->d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *count = v;
    return GBP_OK;
}

int gbp_ba_get_relin_counts(gbp_ba_t *h, int32_t *counts, int32_t n)
{
    ENTER(h);
    if (n < 0 || (n && !counts)) return fail(GBP_EINVAL, "bad argument");
    if (n > RELIN_RING / 2 || n > h->sweep_count)
        return fail(GBP_EINVAL, "only the last min(%d, sweeps run = %ld) sweeps are kept", RELIN_RING / 2, h->sweep_count);
    std::vector<int32_t> ring;
    CHK(download(h, ring, h->d_relin_ring, (size_t)RELIN_RING * RELIN_LANES));
    for (int i = 0; i < n; ++i) {
        const int32_t *w = &ring[(size_t)((h->sweep_count - n + i) % RELIN_RING) * RELIN_LANES];
        int32_t s = 0;
        for (int k = 0; k < RELIN_LANES; ++k) s += w[k];
        counts[i] = s;
    }
    return GBP_OK;
}

int gbp_ba_fill_iters_since_relin(gbp_ba_t *h, int32_t value)
{
    ENTER(h);
    if (value < 0 || value > ITERS_MAX) return fail(GBP_EINVAL, "iters_since_relin %d outside [0, %d]", value, ITERS_MAX);
    const int n = h->p.T * WTILE;
    if (n) hipLaunchKernelGGL(k_fill_iters, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, h->p, n, value);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

// ------------------------------------------------------------------ streaming means export ---
// the reference's viewer thread reads node.mu of every variable once per frame
// (vis/ba_vis.py:35-55).  A snapshot is taken in stream order (between two sweeps) and travels to a pinned host mirror
// on a copy stream, so the sweeps that follow do not wait for PCIe; fetch returns the newest snapshot that has landed.

int gbp_ba_means_snapshot(gbp_ba_t *h)
{
    ENTER(h);
    CHK(peer_check(h, false));
    const Params &p = h->p;
    const size_t n = (size_t)p.C * 6 + (size_t)p.L * 3;
    if (!h->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipEventCreateWithFlags(&h->ev_landed[i], hipEventDisableTiming));
            HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h->h_mu[i]), std::max<size_t>(n, 1) * sizeof(double), hipHostMallocDefault));
        }
        CHK(dev_alloc(h, &h->d_mu, std::max<size_t>(n, 1), false));
    }
    const int b = (int)(h->snap_count & 1);
    if (h->snap_count >= 1) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_landed[(h->snap_count - 1) & 1], 0));   // d_mu is free again
    if (n) hipLaunchKernelGGL(k_pack_means, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, p, h->d_mu);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev_packed, h->stream));
    HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_packed, 0));
    if (n) HIPCHK(hipMemcpyAsync(h->h_mu[b], h->d_mu, n * sizeof(double), hipMemcpyDeviceToHost, h->copy_stream));
    HIPCHK(hipEventRecord(h->ev_landed[b], h->copy_stream));
    h->snap_count++;
    return GBP_OK;
}

int gbp_ba_means_fetch(gbp_ba_t *h, double *cam_mu, double *lmk_mu, int32_t wait)
{
    ENTER(h);
    if (h->snap_count == 0) return fail(GBP_ESTATE, "no snapshot taken yet (gbp_ba_means_snapshot)");
    int b = (int)((h->snap_count - 1) & 1);
    if (wait) {
        HIPCHK(hipEventSynchronize(h->ev_landed[b]));
    } else if (hipEventQuery(h->ev_landed[b]) != hipSuccess) {
        if (h->snap_count < 2) return fail(GBP_ESTATE, "the first snapshot has not landed yet");
        b ^= 1;                                            // the one before it has (copies are issued in order)
        HIPCHK(hipEventSynchronize(h->ev_landed[b]));
    }
    const Params &p = h->p;
    if (cam_mu) std::memcpy(cam_mu, h->h_mu[b], (size_t)p.C * 6 * sizeof(double));
    if (lmk_mu) std::memcpy(lmk_mu, h->h_mu[b] + (size_t)p.C * 6, (size_t)p.L * 3 * sizeof(double));
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

// ------------------------------------------------------------------------ state checkpoint ---
// SURVEY.md 8f rank 4 (the reference keeps its state in Python objects and has no counterpart).  The blob is everything a
// sweep reads or writes -- linearisation points and adaptive variances, both messages, the relinearisation state words,
// beliefs, means and priors -- in the engine's internal order, behind a header that pins the graph it belongs to.

extern "C++" {
namespace {
struct StateHeader {
    char magic[8];                 // "GBPSTATE"
    uint32_t version, has_beliefs;
    uint32_t walk_parity, reserved;    // reserved: bit 0 = the blob carries the dense message remainder
    uint32_t relin_clock, pad;         // the graph's relinearisation clock (the state words hold clock values: gbp_kernels.hpp)
    int32_t F, T, L, C;
    uint64_t graph_hash;           // digest of the factor -> (slot, camera, landmark) maps (k_graph_hash)
    uint64_t payload_bytes;
};

int graph_hash(gbp_ba *h, uint64_t *out)
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

struct StatePart { void *dev; size_t bytes; };

std::vector<StatePart> state_parts(gbp_ba *h)
{
    const Params &p = h->p;
    const size_t S = (size_t)p.T * WTILE;
    return {{p.lin, S * LIN_ROWS * sizeof(double)}, {p.msg, S * MSG_ROWS * sizeof(double)}, {p.avar, p.avar ? S * sizeof(double) : 0},
            {p.lrec, (size_t)p.L * LREC * sizeof(double)}, {p.cbel, (size_t)p.C * CAMREC * sizeof(double)}, {p.cbelief, (size_t)p.C * CBEL * sizeof(double)},
            {p.cprior, (size_t)p.C * 27 * sizeof(double)}, {p.xtra, p.xtra ? S * XTRA_ROW * sizeof(double) : 0}};
}
}  // namespace
}  // extern "C++"

int gbp_ba_state_size(gbp_ba_t *h, uint64_t *bytes)
{
    ENTER(h);
    if (!bytes) return fail(GBP_EINVAL, "bytes is NULL");
    uint64_t n = sizeof(StateHeader);
    for (const StatePart &q : state_parts(h)) n += q.bytes;
    *bytes = n;
    return GBP_OK;
}

int gbp_ba_save_state(gbp_ba_t *h, void *buf, uint64_t bytes)
{
    ENTER(h);
    CHK(peer_check(h, false));
    uint64_t need = 0;
    CHK(gbp_ba_state_size(h, &need));
    if (!buf || bytes < need) return fail(GBP_EINVAL, "state buffer too small: %llu < %llu bytes", (unsigned long long)bytes, (unsigned long long)need);
    StateHeader hd{};
    std::memcpy(hd.magic, "GBPSTATE", 8);
    hd.version = 7; hd.has_beliefs = h->has_beliefs ? 1u : 0u;
    hd.walk_parity = h->walk_parity; hd.reserved = h->p.xtra ? 1u : 0u;      // (1: the payload ends with the dense message remainder)
    hd.relin_clock = (uint32_t)h->p.clk; hd.pad = 0;
    hd.F = h->p.F; hd.T = h->p.T; hd.L = h->p.L; hd.C = h->p.C;
    CHK(graph_hash(h, &hd.graph_hash));
    hd.payload_bytes = need - sizeof(StateHeader);
    char *out = static_cast<char *>(buf);
    std::memcpy(out, &hd, sizeof hd);
    out += sizeof hd;
    for (const StatePart &q : state_parts(h)) {
        if (q.bytes) HIPCHK(hipMemcpyAsync(out, q.dev, q.bytes, hipMemcpyDeviceToHost, h->stream));
        out += q.bytes;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return GBP_OK;
}

// Device-resident checkpoint: the same parts copied to a second set of buffers on the GPU (0.2 GB at 1M factors) and back.
// Restoring costs a device-to-device copy (~0.15 ms at 1M factors) instead of a PCIe upload, and the GPU never idles in between.
int gbp_ba_snapshot_state(gbp_ba_t *h)
{
    ENTER(h);
    std::vector<StatePart> parts = state_parts(h);
    h->snap.resize(parts.size(), nullptr);
    h->snap_bytes.resize(parts.size(), 0);
    for (size_t i = 0; i < parts.size(); ++i) {
        if (h->snap_bytes[i] != parts[i].bytes) {           // (the remainder may have appeared or gone since the last snapshot)
            if (h->snap[i]) { HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(hipFree(h->snap[i])); h->snap[i] = nullptr; }
            if (parts[i].bytes) HIPCHK(hipMalloc(&h->snap[i], parts[i].bytes));
            h->snap_bytes[i] = parts[i].bytes;
        }
        if (parts[i].bytes) HIPCHK(hipMemcpyAsync(h->snap[i], parts[i].dev, parts[i].bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    h->snap_has_beliefs = h->has_beliefs;
    h->snap_parity = h->walk_parity;
    h->snap_clk = h->p.clk;
    return GBP_OK;
}

// a handle whose remainder was switched on on demand goes back to the state "no remainder" (a checkpoint without one is restored)
static int remainder_drop(gbp_ba *h)
{
    if (!h->lazy_xtra || !h->p.xtra) return GBP_OK;
    h->p.xtra = nullptr;
    h->p.crow = CSTAGE_PLAIN;
    h->lazy_xtra = false;
    if (h->fused_suspended) { h->fused.enabled = true; h->dominant = "k_sweep_fused"; }
    return GBP_OK;
}

int gbp_ba_restore_snapshot(gbp_ba_t *h)
{
    ENTER(h);
    h->resid_ok = false;
    if (h->snap.empty()) return fail(GBP_ESTATE, "no snapshot taken (gbp_ba_snapshot_state)");
    const size_t ix = h->snap.size() - 1;                    // the remainder is the last part
    if (h->snap_bytes[ix] && !h->p.xtra) CHK(enable_remainder(h));
    if (!h->snap_bytes[ix] && h->p.xtra) CHK(remainder_drop(h));
    std::vector<StatePart> parts = state_parts(h);
    for (size_t i = 0; i < parts.size(); ++i) {
        if (parts[i].bytes != h->snap_bytes[i]) return fail(GBP_ESTATE, "the snapshot does not fit the handle any more (part %zu)", i);
        if (parts[i].bytes) HIPCHK(hipMemcpyAsync(parts[i].dev, h->snap[i], parts[i].bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    h->has_beliefs = h->snap_has_beliefs;
    h->pending_possible = true;                              // (the restored state words may carry pending relinearisations)
    h->walk_parity = h->snap_parity;
    h->p.clk = h->snap_clk; h->p.clk_inc = 0;
    h->cstage_x0_ok = false;                                // (the staged rows are not part of a checkpoint)
    return GBP_OK;
}

int gbp_ba_load_state(gbp_ba_t *h, const void *buf, uint64_t bytes)
{
    ENTER(h);
    h->resid_ok = false;
    if (!buf || bytes < sizeof(StateHeader)) return fail(GBP_EINVAL, "state buffer too small for a header");
    StateHeader hd;
    std::memcpy(&hd, buf, sizeof hd);
    if (std::memcmp(hd.magic, "GBPSTATE", 8) != 0) return fail(GBP_EINVAL, "not a GBP state blob (magic)");
    if (hd.version != 7)            // 1-3: dense / core-only message layouts, 4: beliefs without covariances, 5: state / meta words in arrays of their own, 6: iters_since_relin stored instead of clock values
        return fail(GBP_EINVAL, "unsupported state blob version %u (this library reads and writes version 7; INTEGRATION.md)", hd.version);
    uint64_t mine = 0;
    CHK(graph_hash(h, &mine));
    if (hd.F != h->p.F || hd.T != h->p.T || hd.L != h->p.L || hd.C != h->p.C || hd.graph_hash != mine)
        return fail(GBP_EINVAL, "state blob belongs to a different graph (F/L/C or factor order differ)");
    // Everything is validated BEFORE the handle is touched: the blob decides whether the handle carries a dense remainder, and a rejected
    // blob must leave the handle as it was.
    const bool blob_xtra = (hd.reserved & 1u) != 0;
    if (!blob_xtra && h->p.xtra && !h->lazy_xtra)
        return fail(GBP_EINVAL, "the state blob has no dense message remainder but this graph always carries one (num_undamped_iters = 0)");
    uint64_t need = sizeof(StateHeader);
    {
        const std::vector<StatePart> parts = state_parts(h);                      // (the remainder is the last part)
        for (size_t i = 0; i + 1 < parts.size(); ++i) need += parts[i].bytes;
        if (blob_xtra) need += (size_t)h->p.T * WTILE * XTRA_ROW * sizeof(double);
    }
    if (bytes < need || hd.payload_bytes != need - sizeof(StateHeader)) return fail(GBP_EINVAL, "state blob truncated");
    if (blob_xtra && !h->p.xtra) CHK(enable_remainder(h));
    if (!blob_xtra && h->p.xtra) CHK(remainder_drop(h));
    const char *in = static_cast<const char *>(buf) + sizeof hd;
    for (const StatePart &q : state_parts(h)) {
        if (q.bytes) HIPCHK(hipMemcpyAsync(q.dev, in, q.bytes, hipMemcpyHostToDevice, h->stream));
        in += q.bytes;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->has_beliefs = hd.has_beliefs != 0;
    h->pending_possible = true;                              // (the loaded state words may carry pending relinearisations)
    h->walk_parity = hd.walk_parity & 1u;
    h->p.clk = (int)(hd.relin_clock & CLK_MASK); h->p.clk_inc = 0;
    h->cstage_x0_ok = false;
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

int gbp_ba_comm_info(gbp_ba_t *h, int32_t *kind, int32_t *rank, int32_t *n_ranks)
{
    ENTER(h);
    int k = GBP_COMM_NONE, n = h->xch_ranks;
    if (h->peer.connected) { k = GBP_COMM_PEER; n = h->peer.n_ranks; }
    else if (h->comm) {
        k = GBP_COMM_RCCL;
        if (g_rccl.CommCount) {
            int cnt = 0;
            const ncclResult_t rc = g_rccl.CommCount(h->comm, &cnt);
            if (rc != ncclSuccess) return fail(GBP_EHIP, "ncclCommCount failed: %s", g_rccl.GetErrorString(rc));
            n = cnt;                                         // what RCCL itself says
        }
    } else if (h->xch_fn) k = GBP_COMM_CALLBACK;
    if (kind) *kind = k;
    if (rank) *rank = h->xch_rank;
    if (n_ranks) *n_ranks = n;
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

int gbp_ba_eval_fn(const double *K4, int32_t n, const double *x9, double *h2, double *J18, double *hproj2, int32_t device)
{
    if (!K4 || n < 0 || (n && !x9)) return fail(GBP_EINVAL, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GBP_ENODEV, "no HIP device visible: libgbp_hip.so has no CPU path");
    if (device < 0 || device >= ndev) return fail(GBP_EINVAL, "device %d out of range (%d visible)", device, ndev);
    if (!n) return GBP_OK;
    HIPCHK(hipSetDevice(device));
    double *d = nullptr;
    const size_t N = (size_t)n;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d), sizeof(double) * N * (9 + 2 + 18 + 2)));
    double *d_x = d, *d_h = d + 9 * N, *d_J = d_h + 2 * N, *d_hp = d_J + 18 * N;
    hipError_t e = hipMemcpy(d_x, x9, sizeof(double) * 9 * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_eval_fn, dim3(grid_for(N)), dim3(BLOCK), 0, nullptr, Intrinsics{K4[0], K4[1], K4[2], K4[3]}, n, d_x, d_h, d_J, d_hp);
        e = hipGetLastError();
    }
    if (e == hipSuccess && h2) e = hipMemcpy(h2, d_h, sizeof(double) * 2 * N, hipMemcpyDeviceToHost);
    if (e == hipSuccess && J18) e = hipMemcpy(J18, d_J, sizeof(double) * 18 * N, hipMemcpyDeviceToHost);
    if (e == hipSuccess && hproj2) e = hipMemcpy(hproj2, d_hp, sizeof(double) * 2 * N, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(GBP_EHIP, "gbp_ba_eval_fn: %s", hipGetErrorString(e));
    return GBP_OK;
}

// phase profile of the last fused sweep (GBP_PHASE_TIMING builds; rows = workgroups x waves, NPHASE columns of s_memtime ticks)
int gbp_ba_phase_profile(gbp_ba_t *h, uint64_t *out, int32_t cap_rows, int32_t *n_rows, int32_t *n_cols)
{
    ENTER(h);
    if (n_rows) *n_rows = 0;
    if (n_cols) *n_cols = NPHASE;
    if (!h->fused.enabled || !h->fused.args.phase) return fail(GBP_ESTATE, "not a GBP_PHASE_TIMING build (tools/phase_profile.py)");
    const int rows = h->fused.n_blocks * WAT_WAVES;
    if (n_rows) *n_rows = rows;
    if (out && cap_rows >= rows) {
        HIPCHK(hipMemcpyAsync(out, h->fused.args.phase, sizeof(uint64_t) * (size_t)rows * NPHASE, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
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
        (int32_t)h->big_lmks.size()};
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
