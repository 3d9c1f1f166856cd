// gbp_capi_state.hip -- libgbp_hip.so, checkpoints: the host blob (gbp_ba_save_state / gbp_ba_load_state) and the device-resident slot
// (gbp_ba_snapshot_state / gbp_ba_restore_snapshot).  Copies only: no kernel is launched from this unit.
#include "gbp_handle.hpp"

extern "C" {

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


}  // extern "C"
