// gbp_handle.hpp -- what the translation units of libgbp_hip.so share about a handle (private: include/gbp_ba.h is the boundary).
//
//   gbp_capi.hip        life cycle: errors, create / destroy (graph build), streams, priors, BAL reader, plan info, layout checks
//   gbp_capi_sweep.hip  every launch of a sweep: fused + general + stage-wise, the dense remainder, diagnostics, kernel timing
//   gbp_capi_shard.hip  the landmark-sharded loop: RCCL, peer-store mailboxes, the exchange between reduce and finish
//   gbp_capi_views.hip  state views (beliefs, messages, factors, relinearisation state), streaming means, eval_fn
//   gbp_capi_state.hip  checkpoints (host blob, device slot)
//
// Kernels live with the unit that launches them (a __global__ defined in a header may be instantiated by one unit only: the
// dynamic-LDS attributes of the fused sweep are set on the very function objects that are launched).
#pragma once
#include "../../include/gbp_ba.h"
#include "gbp_kernels.hpp"
#include "gbp_fused_plan.hpp"

#include <rccl/rccl.h>      // types only: the library is dlopen()ed when a communicator is asked for (no link-time dependency)

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace gbp {
int set_error(int code, const char *fmt, ...);      // gbp_capi.hip: the thread-local message behind gbp_last_error(); returns `code`
}
#define fail(...) ::gbp::set_error(__VA_ARGS__)

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(e__ == hipErrorOutOfMemory ? GBP_ENOMEM : GBP_EHIP, "%s failed: %s (%s:%d)",   \
                        #expr, hipGetErrorString(e__), __FILE__, __LINE__);                            \
    } while (0)

#define CHK(expr) do { int rc__ = (expr); if (rc__ != GBP_OK) return rc__; } while (0)

#define ENTER(h)                                                  \
    if (!(h)) return fail(GBP_EINVAL, "null handle");             \
    HIPCHK(hipSetDevice((h)->device))

using namespace gbp;

struct gbp_ba {
    Params p{};
    int device = 0;
    int flags = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // order maps: on the device (built there, gbp_build.hpp); the host keeps only what is L- or C-sized
    int *d_ref_cam = nullptr, *d_ref_lmk = nullptr;   // per reference factor (p.cadj = reference id -> slot, p.cpos = slot -> reference id)
    int pack_mode = 0, n_big = 0;                // tile packing (build_graph): 0 whole landmarks, 1 + chunk tiles of the n_big landmarks above 64 factors, 2 dense
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
    std::vector<int4> wg_win;                    // camera windows, per workgroup of the fused sweep: {lowest camera, cameras in its set, offset into wg_cams, width of
                                                 // the interval lowest .. highest}; empty: no windows (build_graph decides, fused_plan consumes)
    std::vector<int> wg_cams;                    // the workgroups' camera sets, ascending, one after the other
    bool staged_auto = false;                    // the general sweep was picked by the sparseness rule (build_graph), not asked for
    int staged_xchg_blocks[3] = {0, 0, 0};       // grid cap of k_cam_staged_xchg<64 | 128 | 256> on this device (0: not asked yet)
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

// ---- small helpers every unit uses (inline: one definition per unit, no state of their own) ----------------------------------

template <typename T>
inline int dev_alloc(gbp_ba *h, T **out, size_t n, bool zero = true)
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
inline int arena_reserve(gbp_ba *h, size_t bytes)
{
    if (h->arena || getenv("GBP_NO_ARENA")) return GBP_OK;
    HIPCHK(hipMalloc(&h->arena, bytes));
    h->allocs.push_back(h->arena);
    h->arena_bytes = bytes;
    h->arena_used = 0;
    return GBP_OK;
}

inline void *arena_take(void *ctx, size_t bytes)          // FusedPlan's allocator hook
{
    gbp_ba *h = static_cast<gbp_ba *>(ctx);
    const size_t off = (h->arena_used + 4095) & ~(size_t)4095;
    if (!h->arena || off + bytes > h->arena_bytes) return nullptr;
    h->arena_used = off + bytes;
    return static_cast<char *>(h->arena) + off;
}

inline int ensure_tmp(gbp_ba *h, size_t bytes)
{
    if (bytes <= h->tmp_bytes) return GBP_OK;
    if (h->d_tmp) { HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(hipFree(h->d_tmp)); h->d_tmp = nullptr; h->tmp_bytes = 0; }
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->d_tmp), bytes));
    h->tmp_bytes = bytes;
    return GBP_OK;
}

template <typename T>
inline int upload(gbp_ba *h, T *dst, const std::vector<T> &src)
{
    if (src.empty()) return GBP_OK;
    HIPCHK(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));      // src is a temporary
    return GBP_OK;
}

template <typename T>
inline int download(gbp_ba *h, std::vector<T> &dst, const T *src, size_t n)
{
    dst.resize(n);
    if (!n) return GBP_OK;
    HIPCHK(hipMemcpyAsync(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return GBP_OK;
}

inline int grid_for(size_t n) { return (int)((n + BLOCK - 1) / BLOCK); }

// The relinearisation clock (gbp_kernels.hpp, state word): every call in which the reference's relinearise_factors() runs -- a sweep
// with local_relin, or the stage call itself -- advances it by one; a factor's iters_since_relin is the clock minus the value its
// state word holds.  Set before the launch: the kernels read the value AFTER the call's advance (Params::clk) and whether it advanced.
inline void clock_tick(gbp_ba *h, bool advance)
{
    h->p.clk_inc = advance ? 1 : 0;
    if (advance) h->p.clk = (int)(((unsigned)h->p.clk + 1u) & CLK_MASK);
}


inline size_t n_slots(const gbp_ba *h) { return std::max<size_t>((size_t)h->p.T * WTILE, 1); }

// mailbox geometry: [2 halves][n_ranks][C] rows of PEER_ROW doubles (27 sums | tag), then [n_ranks] probe rows (gbp_ba_peer_selftest)
inline size_t peer_block(const gbp_ba *h) { return (size_t)std::max(h->p.C, 1) * PEER_ROW; }
inline size_t peer_bytes(const gbp_ba *h, int n) { return (2 * (size_t)n * peer_block(h) + (size_t)n * PEER_ROW) * sizeof(double); }   // + the self-test's probe rows
inline double *peer_probe(const gbp_ba *h, void *base, int n, int src) { return static_cast<double *>(base) + 2 * (size_t)n * peer_block(h) + (size_t)src * PEER_ROW; }
inline double *peer_data(const gbp_ba *h, void *base, int n, int half, int src)
{
    return static_cast<double *>(base) + ((size_t)half * n + src) * peer_block(h);
}


// ---- functions one unit defines and others call ----------------------------------------------------------------------------------
namespace gbp {
// gbp_capi.hip
int peer_check(gbp_ba *h, bool clear);               // a finish wave of the peer-store exchange gave up waiting: GBP_ESTATE until gbp_ba_sync has reported it
int graph_hash(gbp_ba *h, uint64_t *out);            // digest of the factor -> (slot, camera, landmark) maps (state blobs)
// gbp_capi_sweep.hip
int plan_fused_sweep(gbp_ba *h, int n_cus);          // fused_plan on the handle (the kernels whose attributes it sets live in that unit)
int fused_max_cams_of_this_build();
int sweep_begin(gbp_ba *h, int with_messages, int robustify, int local_relin, double *partial, int finish = 0, bool *finished = nullptr,
                bool defer_big = false, const PeerOut *peer = nullptr, const PeerWait *merged = nullptr);
int launch_cam_finish(gbp_ba *h, const double *gathered, int n_parts, size_t stride, const PeerWait *wait = nullptr);
int launch_finish_parts(gbp_ba *h, hipStream_t stream);      // beliefs of the landmarks that span tiles (after a sweep's factor kernel)
int launch_peer_selftest(gbp_ba *h, const PeerOut &po, double *mine, int rank, long long ticks, int *d_out);
int enable_remainder(gbp_ba *h);
int remainder_drop(gbp_ba *h);
int remainder_guard(gbp_ba *h, int local_relin, int no_test);
int remainder_release(gbp_ba *h);
// gbp_capi_shard.hip
void peer_release(gbp_ba *h);
void shard_comm_release(gbp_ba *h);
}  // namespace gbp
