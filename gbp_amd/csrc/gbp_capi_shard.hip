// gbp_capi_shard.hip -- libgbp_hip.so, the landmark-sharded loop (no reference counterpart): the exchange of the camera
// partial sums between a rank's reduce and its finish -- an RCCL all-gather on the library's own communicator, the peer-store
// mailboxes, or a caller's function -- and gbp_ba_iterate_sharded around it.  No kernel is launched from this unit directly: the
// sweep's launches are gbp_capi_sweep.hip's (gbp_handle.hpp).
#include "gbp_handle.hpp"

#include <dlfcn.h>

#include <mutex>

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

void gbp::peer_release(gbp_ba *h)
{
    gbp_ba::Peer &pe = h->peer;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int r = 0; r < MAX_PEERS; ++r) {
        if (pe.opened[r] && pe.base[r]) (void)hipIpcCloseMemHandle(pe.base[r]);
        pe.opened[r] = false; pe.base[r] = nullptr;
    }
    pe.connected = false;
}

void gbp::shard_comm_release(gbp_ba *h)
{
    if (h->comm && g_rccl.CommDestroy) { (void)hipStreamSynchronize(h->stream); (void)g_rccl.CommDestroy(h->comm); }
    h->comm = nullptr;
    if (h->xch_fn == rccl_exchange) { h->xch_fn = nullptr; h->xch_ctx = nullptr; h->xch_ranks = 1; h->xch_rank = 0; }
}


extern "C" {

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
    // (test switch: a node whose RCCL cannot come up -- the callers' fallbacks, ShardedBA and bench.py's line, are exercised with it)
    if (getenv("GBP_RCCL_FAIL")) return fail(GBP_ESTATE, "RCCL switched off by GBP_RCCL_FAIL (test switch)");
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
    HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(pe.mailbox), (int)PEER_EMPTY32, bytes / 4, h->stream));      // every slot empty (gbp_kernels.hpp: the data is its own arrival flag)
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
    po.n = n; po.C = h->p.C; po.stale = nullptr; po.seq = 0x9b50000000000000ull | ++pe.probe_seq;
    for (int r = 0; r < n; ++r) po.dst[r] = peer_probe(h, pe.base[r], n, pe.rank);
    int clk_khz = 0;
    if (hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || clk_khz <= 0) clk_khz = 100000;
    const long long ticks = (long long)((double)std::max(1, timeout_ms) * (double)clk_khz);
    HIPCHK(hipMemsetAsync(pe.d_ctl + 2, 0, 2 * sizeof(int), h->stream));
    CHK(launch_peer_selftest(h, po, peer_probe(h, pe.mailbox, n, 0), pe.rank, ticks, pe.d_ctl + 2));
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
    po.n = n; po.C = h->p.C; po.seq = pe.seq;
    po.stale = peer_data(h, pe.mailbox, n, half ^ 1, 0);      // what the previous exchange delivered here: read then, emptied by this push
    for (int r = 0; r < n; ++r) po.dst[r] = peer_data(h, pe.base[r], n, half, pe.rank);
    PeerWait w{peer_data(h, pe.mailbox, n, half, 0), pe.seq, pe.timeout_ticks, pe.d_ctl + 1, nullptr};
    // Without a rendezvous hook everything behind the fused sweep is ONE launch (k_cam_reduce_xchg); logical ranks on one device
    // (the hook is set) keep reduce / push and finish apart, with the hook between them, so that they never spin on each other.
    const bool merged = !h->xch_fn && with_messages && !getenv("GBP_PEER_SPLIT");      // (fused: k_cam_reduce_xchg; general: k_cam_staged_xchg)
    bool finished = false;
    // The beliefs of the landmarks that span tiles (k_lmk_finish_parts) go out on THIS stream, as a short launch between the sweep
    // kernel and the reduce (1.3 us, EXPERIMENTS.md round 5).  Round 5 forked them onto a side stream "beside the exchange" -- but the
    // fork event was recorded behind the merged reduce -> push -> wait -> finish launch, so they ran strictly after it, plus two event
    // hops.  Here the exchange IS that one launch: there is nothing to run beside.
    CHK(sweep_begin(h, with_messages, robustify, local_relin, h->d_send, 0, &finished, false, &po, merged ? &w : nullptr));
    if (!finished) {
        if (h->xch_fn) {                                     // rendezvous hook (logical ranks on ONE device: tests)
            int rc = h->xch_fn(h->xch_ctx, nullptr, nullptr, 0, h->stream);
            if (rc != GBP_OK) return rc < 0 ? rc : fail(GBP_EHIP, "the rendezvous function returned %d", rc);
        }
        CHK(launch_cam_finish(h, nullptr, n, 0, &w));
    }
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
    // landmarks that span tiles: their beliefs need nothing from the exchange.  The side stream picks them up behind the reduce launch
    // (the fork event follows it) and runs them while the exchange function -- an RCCL all-gather, a caller's MPI -- owns this stream.
    const bool big = with_messages && h->p.parts != nullptr;
    CHK(sweep_begin(h, with_messages, robustify, local_relin, h->d_send, 0, nullptr, big));
    if (big) {
        if (!h->side_stream) {
            HIPCHK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
        CHK(launch_finish_parts(h, h->side_stream));
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


}  // extern "C"
