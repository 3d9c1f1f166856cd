// gbp_fused.hpp -- the whole synchronous_iteration (gbp.py:86-92) except the camera solve, as ONE kernel.
//
// Why: the general sweep re-reads every message after the barrier (variable stage) and gathers the camera
// messages through an index list (8-byte gathers = 8x read amplification).  Here each message is read once
// and written once per sweep, and everything else stays on chip:
//
//   * a TILE (64 consecutive factor slots = one wavefront, see gbp_kernels.hpp) owns its landmarks
//     completely, so the new landmark messages go through the wave's LDS scratch and the landmark beliefs
//     (prior + sum in adj_factors order, gbp.py:182-188) are formed by the same wave;
//   * camera messages are accumulated into a per-workgroup LDS table acc[C][27] (500 cameras = 108 KB of
//     the CU's 160 KB).  One workgroup per CU walks a fixed contiguous range of tiles and stores its table
//     at the end ([camera][workgroup][27]); k_cam_reduce_tree sums the tables per camera in a fixed order and,
//     on a single GPU, finishes the camera belief (prior + sum, 6x6 solve) in the same launch;
//   * the 8 waves of a workgroup are autonomous: each pulls the next tile of the workgroup's range from an
//     LDS counter and never meets the others at an s_barrier, so while one wave waits for HBM another does
//     fp64 maths on the same SIMD (two waves per SIMD, 256 VGPRs each);
//   * the landmark-belief phase of a tile is pure LDS work; it runs one iteration LATE, after the next tile's
//     loads have been issued, so the wave keeps HBM requests in flight (139 -> 116 us per sweep at 1M factors);
//   * determinism without barriers: a wave may add its tile's camera messages to acc only when all earlier
//     tiles of the workgroup have done so (LDS ticket `done`), and lanes of one tile that hit the same camera
//     add in the order of a pre-computed rank (kept in the state word).  The summation order is therefore
//     (workgroup, tile, rank) -- independent of which wave ran which tile and of timing.  Within a rank round
//     every lane targets a different camera, so the LDS ds_add_f64 is a plain read-modify-write.
//
// HBM traffic per sweep (bench.py layout_bytes): F*(21 read + 10 written doubles + 8 B of meta | state words)
// + L*(a 20-double record read + 9 doubles of mean | covariance written) + the workgroup tables (256 * C rows of 28 doubles,
// written here and read once by the reduce) -- under a third of the "algorithmic" 1072 B per factor of SURVEY.md 8d, which
// assumed dense messages and a second pass over them.
//
// Tried and measured on MI355X, not kept (round 2): a software-pipelined loop that requests everything tile k+1 reads (streams,
// landmark records, camera records) during the back half of tile k.  With no spills and the operands of every tile already on
// chip when its maths starts ("wait for loads" 1.4 % of the wave-time, tools/phase_profile.py) the sweep was SLOWER, 108 us
// against 84 us: the time moved into the issue of the stores and the in-order accumulation wait (23 % + 25 %) -- the vector-memory
// path of the CU (camera gathers: 17 x 64 different lines per tile, streams, stores) and HBM are the limit, not latency.
//
// Also measured and not kept: knowing a wave's NEXT tile one iteration early (ticket, descriptor and meta word fetched during
// the current tile, so that streams and camera gather leave in one go instead of after two dependent round trips): 89 us with the
// lookahead alone, 92-94 us with the gather issued up-front (256 VGPRs, one spill) against 81 us.  A wave that commits to its next
// tile early takes away what the ticket counter is for -- whichever wave is free takes the next tile -- and every other wave of
// the workgroup then waits for it at the in-order accumulation.
//
// The camera accumulation is the most expensive thing left on chip (round 2, same box, alternating runs): doing it TWICE (halves)
// costs +22 us of 82; only its first round (GBP_FUSED_DBG=32: the ~6 % of a tile's factors that share a camera with an earlier
// lane are dropped) saves 6 us; the same adds as explicit ds_read / v_add / ds_write instead of ds_add_f64 -- legal, since one
// wave at a time adds and the lanes of a round hit different cameras -- cost +9 us (the read's round trip lands inside the
// ordered section, the no-return atomics only have to be issued).  64 random cameras per instruction fall on 32 bank pairs, about
// three times the conflict-free LDS time; a transposed accumulation (27 lanes per factor, consecutive banks) would need the 27
// values of a factor in LDS first, and there is no LDS left next to the table (500 cameras: 156 of 160 KB).  Splitting the
// ordered section into three (nine entries each, a turn counter per third, so that three waves can be inside): 84.7 against
// 79.5 us per step -- the ordered section is not what the waves queue for, two more hand-overs per tile only cost.
//
// Landmarks that span tiles (more than 64 factors: chunk tiles; or the dense packing, gbp_kernels.hpp header): every tile adds up its
// part of their messages, and their beliefs are formed afterwards by k_lmk_finish_parts.  If acc + the per-wave scratch do not fit the LDS
// (C > 516) the cameras are split into two groups (516 + up to 758): the sweep adds up the first, k_cam_pass the others; beyond that the
// plan stays disabled and the general sweep runs.
#pragma once
#include "gbp_kernels.hpp"
#include "gbp_fused_plan.hpp"
#include <mutex>

namespace gbp {

// Instrumented launches: workgroup 0 stores the clock when it starts (a grid starts first -> last within ~0.5 us).  One plain store:
// an atomicMin / atomicMax from every workgroup on one word costs ~12 ns each at the memory side -- 500 of them made the reduce
// launch 3 us longer -- and waiting for a whole workgroup at the end of a short launch cost more still.  Consecutive start stamps
// tile the stream's timeline the way rocprofv3's kernel durations do.
GBP_DEV void clk_begin(unsigned long long *clk)
{
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) *clk = (unsigned long long)wall_clock64();
}

// Accesses of the persistent loop: a wave-uniform base pointer (SGPR pair) + an unsigned 32-bit lane offset, so that the address of
// a load or store costs ONE vector register (global_load ... v_off, s[base:base+1]) instead of a 64-bit pair per access -- with
// per-access 64-bit addresses the loop kept ~30 address registers alive (or spilled them) between a tile's loads and its stores.
// (Nontemporal loads / stores for the streams were measured in round 3 and lost: +12...18 us, they bypass the memory-side cache.)
// (the offset is in BYTES and 32 bits wide: base + zext(offset) is the one address shape the saddr form of global_load takes)
GBP_DEV double2 ld2(const double *__restrict__ base, unsigned byte_off)
{
    return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + byte_off);
}
GBP_DEV void st2(double *__restrict__ base, unsigned byte_off, double x, double y)
{
    *reinterpret_cast<double2 *>(reinterpret_cast<char *>(base) + byte_off) = make_double2(x, y);
}
GBP_DEV void st1(double *__restrict__ base, unsigned byte_off, double x) { *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byte_off) = x; }
GBP_DEV void st2_nt(double *__restrict__ base, unsigned byte_off, double x, double y)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d v; v.x = x; v.y = y;
    __builtin_nontemporal_store(v, reinterpret_cast<v2d *>(reinterpret_cast<char *>(base) + byte_off));
}
// (nts is wave-uniform: a scalar branch)
GBP_DEV void st2x(double *__restrict__ base, unsigned byte_off, double x, double y, bool nts)
{
    if (nts) st2_nt(base, byte_off, x, y); else st2(base, byte_off, x, y);
}

// everything a tile's factors stream: six + five row pairs of the tile's block (16 bytes per lane each), the meta and state words
struct TileStreams {
    double2 a[6], m[5];       // a[5] = z[1] | {meta, state}
    double avar;              // robust losses only
    int cpos;
};
GBP_DEV double2 ld2_nt(const double *__restrict__ base, unsigned byte_off)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(reinterpret_cast<const char *>(base) + byte_off));
    return make_double2(v.x, v.y);
}

// L2 TOUCH-PREFETCH (round 6).  The sweep takes 65 us where its memory traffic alone takes 44 and its arithmetic 27: a wave has loads
// in flight only between the top of an iteration and the arrival of its streams, and none while it does its maths -- eight waves per
// CU are then too few requests in flight to keep the memory system busy.  Loading the NEXT tile's streams into registers during the
// maths was built three times (rounds 2, 3, 4: 84-108 us: 44 more live registers, the stores queue behind eleven more 1 KB loads).
// This needs neither: after its own stream loads a wave touches one byte of every 64 bytes of the tile that will be taken
// GBP_PF_DIST tickets later (three instructions, the loaded bytes are thrown away), so that the 11 KB block is on its way into the
// XCD's L2 -- 4 MB, 128 KB per CU: eight tiles ahead is 88 KB -- while this wave and its neighbours compute; whichever wave draws
// that ticket finds its streams one L2 hit away instead of one trip to the memory side.
#ifndef GBP_PF_DIST
#define GBP_PF_DIST 0
#endif
#ifndef GBP_PF_STRIDE
#define GBP_PF_STRIDE 64
#endif
struct Touch { unsigned v[3]; };
template <bool NT>
GBP_DEV void touch_tile(const Params &p, int t, int lane, Touch &o)
{
    constexpr int LIN_B = LIN_ROWS * WTILE * 8, MSG_B = MSG_ROWS * WTILE * 8, N_LIN = LIN_B / GBP_PF_STRIDE, N = (LIN_B + MSG_B) / GBP_PF_STRIDE;
    const char *lin_t = reinterpret_cast<const char *>(p.lin + (size_t)t * (LIN_ROWS * WTILE));
    const char *msg_t = reinterpret_cast<const char *>(p.msg + (size_t)t * (MSG_ROWS * WTILE));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = j * 64 + lane;
        o.v[j] = 0;
        if (j * 64 < N && i < N) {
            const unsigned char *q = reinterpret_cast<const unsigned char *>(i < N_LIN ? lin_t + i * GBP_PF_STRIDE : msg_t + (i - N_LIN) * GBP_PF_STRIDE);
            o.v[j] = NT ? __builtin_nontemporal_load(q) : *q;
        }
    }
}
// the loaded bytes must be waited for somewhere (their registers are dead otherwise and would be handed out while the loads are in flight)
GBP_DEV void touch_retire(const Touch &o) { asm volatile("" ::"v"(o.v[0]), "v"(o.v[1]), "v"(o.v[2])); }

// nt: bit 0 = the lin rows (x0 | z | variance) stream PAST the memory-side cache, bit 1 = the message rows too (nontemporal loads:
// no allocation in the 256 MiB Infinity Cache).  The fused sweep of a graph whose whole working set fits that cache uses neither (at the
// headline size bypassing it costs +12...18 us); a larger graph sends the tiles that do not fit past it, loads and stores, so that the
// rest STAYS resident from sweep to sweep instead of everything thrashing (FusedArgs::pin, fused_plan); the general sweep sends both past it
// so that the cache keeps the staged camera rows for k_cam_partial_staged (126 against 132 us per sweep at 1M factors).
template <int LOSS, bool STAGED>
GBP_DEV void issue_streams(const Params &p, int t, int lane, TileStreams &s, int nt)
{
    if (LOSS != 0) s.avar = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(p.avar + (size_t)t * WTILE) + (unsigned)lane * 8u);
    if (STAGED) s.cpos = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(p.cpos + (size_t)t * WTILE) + (unsigned)lane * 4u);
    const double *lin_t = p.lin + (size_t)t * (LIN_ROWS * WTILE);
    const double *msg_t = p.msg + (size_t)t * (MSG_ROWS * WTILE);
    const unsigned lo = (unsigned)lane * 16u;             // byte offset of this lane's 16 bytes inside a row pair (1 KB per pair)
    if (nt & 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s.a[k] = ld2_nt(lin_t, 1024u * k + lo);
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) s.a[k] = ld2(lin_t, 1024u * k + lo);
    }
    if (nt & 2) {
#pragma unroll
        for (int k = 0; k < 5; ++k) s.m[k] = ld2_nt(msg_t, 1024u * k + lo);
    } else {
#pragma unroll
        for (int k = 0; k < 5; ++k) s.m[k] = ld2(msg_t, 1024u * k + lo);
    }
}

// STAGED = the general sweep (any number of cameras): instead of adding its camera messages into the workgroup's LDS table a tile
// writes what rebuilds them (x0 9 | q_C 2 | W 3 in one whole 128-byte line per factor) to cstage[cpos[slot]], i.e. in the camera's own
// adj_factors order, and k_cam_partial_staged reads one contiguous run per camera.  No table, no ordered section; everything else --
// the persistent loop, the late landmark beliefs, the addressing -- is shared with the fused sweep.  (Rounds 1-3 ran the general sweep
// as one wave per tile, k_factor_tile: 107 us at 1M factors; that kernel now serves the stage-wise calls and the dense remainder.)
constexpr int STAGED_WAVE_DOUBLES = WAVE_LDS_DOUBLES + WTILE * CSTAGE_PLAIN + WTILE / 2;      // messages | rows | cpos
// WINDOWED: the workgroup's LDS table covers only the cameras its own tiles meet (FusedArgs::win, wgcams) -- sequences, where a landmark
// is seen by neighbouring cameras and the landmarks are numbered along the trajectory: any number of cameras, tables that shrink with
// the windows (fused_plan).  A 16-bit map behind the control words turns (camera - lowest camera of the set) into the table row.
template <int LOSS, int NWAVES, bool STAGED = false, bool PINNED = false, bool SINGLE = false, bool WINDOWED = false>      // PINNED: FusedArgs::pin is in force (graphs beyond the memory-side cache); SINGLE: see the accumulation
__global__ __launch_bounds__(NWAVES * 64) void k_sweep_wat(Params p, FusedArgs a, const int4 *__restrict__ tiles)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *acc = smem;                                              // [C][27]
    const int acc_even = (a.acc_doubles + 1) & ~1, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PER_WAVE = STAGED ? STAGED_WAVE_DOUBLES : WAVE_LDS_DOUBLES;
    double *wl = smem + acc_even + wave * PER_WAVE;
    double *wr = wl + WAVE_LDS_DOUBLES;                              // STAGED: [64][16] rows on their way to cstage
    int *wpos = reinterpret_cast<int *>(wr + WTILE * CSTAGE_PLAIN);  // STAGED: their row numbers
    int *ctl = reinterpret_cast<int *>(smem + acc_even + NWAVES * PER_WAVE);   // {next, done}
    const int tid = threadIdx.x, lane = tid & 63;
    clk_begin(a.clk);
    int cam_base = a.cam_base, cam_count = a.cam_count, row_off = 0;
    if (WINDOWED) {
        const int4 w = a.win[blockIdx.x];
        cam_base = __builtin_amdgcn_readfirstlane(w.x); cam_count = __builtin_amdgcn_readfirstlane(w.y); row_off = __builtin_amdgcn_readfirstlane(w.z);
        // behind the control words: where table row k goes (rowidx, fetched NOW: read at the write-out it was a dependent global load in
        // front of every 16-byte store, after everything else had finished), then the 16-bit map camera -> table row
        int *rowl = ctl + 2;
        unsigned short *map = reinterpret_cast<unsigned short *>(rowl + win_rows_ints(a.acc_doubles / 27));
        for (int i = tid; i < cam_count; i += NWAVES * 64) {
            map[a.wgcams[row_off + i] - cam_base] = (unsigned short)i;
            rowl[i] = a.rowidx[row_off + i];
        }
    }
    for (int i = tid; i < a.acc_doubles; i += NWAVES * 64) acc[i] = 0.0;
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; }
    __syncthreads();
    // the workgroup's contiguous range of tiles, computed (a table of range starts cost a dependent load before the first tile)
    // STRIDED WALK beyond the memory-side cache (round 6).  With a contiguous range per workgroup the sweep reads 256 streams that lie
    // megabytes apart, and how evenly those fall on the HBM channels depends on where the driver happened to put the pages: the same
    // binary ran 6 % faster or slower from process to process -- from engine to engine inside one process -- ("two modes",
    // EXPERIMENTS.md rounds 5-6; a plain copy of the same memory does not see it).  Workgroup b walking tiles b, b + n, b + 2 n, ...
    // keeps all 256 streams inside one moving window of a few megabytes: every one of sixteen engines behind different amounts of other
    // memory lands at or below the old fast mode (2M factors 158.5-161.0 us per step against 165.0-171.1, 10M 698 against 722).  The tiles
    // a workgroup keeps cacheable (FusedArgs::pin) are then one contiguous piece of the graph.  Not with camera windows: a workgroup's
    // camera set lives on its tiles being neighbours.  Not below the cache size: there the contiguous walk is as fast (round 5: 74.6
    // against 74.3 us) and the plain kernel stays as it is.  (-DGBP_CONTIGUOUS_PINNED: the old walk everywhere, for A/B runs.)
#if defined(GBP_CONTIGUOUS_PINNED)
    constexpr bool STRIDED = false;
#else
    // (the general sweep -- STAGED: every stream nontemporal, the working set far beyond the cache at any size that matters -- the same:
    //  117.7-119.8 -> 113.6-113.9 us per sweep at 1M factors x 500 cameras, 123.9-125.1 -> 118.4-118.9 at 2 000; its camera sums are made in
    //  the cameras' own order by another kernel, so its results do not change by a bit)
    constexpr bool STRIDED = (PINNED && !WINDOWED) || STAGED;
#endif
    const int tb = STRIDED ? 0 : (int)((long long)blockIdx.x * p.T / gridDim.x);
    const int ntl = STRIDED ? (p.T - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : (int)((long long)(blockIdx.x + 1) * p.T / gridDim.x) - tb;

    // The landmark beliefs of a tile (LDS work, no loads) are formed one iteration LATE, after the next tile's loads have
    // been issued, so that the wave has HBM requests in flight meanwhile.  The tile's landmark messages wait in the wave's LDS
    // scratch, which the next tile overwrites only afterwards; the priors they are added to (one entry per lane and pass) and the
    // landmarks' slot ranges are fetched with the tile's own streams and wait in nine registers.
    bool pend = false;
    int q_t = 0, q_l0 = 0, q_nl = 0;
    LmkPre pre;
#pragma unroll
    for (int b = 0; b < LMK_PASSES; ++b) pre.pri[b] = 0.0;
    pre.rx = 0; pre.ry = 0;
    int n_relin = 0;                                      // factors of this wave's tiles that relinearised (wave-uniform)
    GBP_PH_DECL;

    const int lane_entry = lane;
    TileStreams S;
    for (;;) {
        GBP_PH_NOWAIT(9);                                  // loop overhead / after the release of the accumulation ticket
        // Everything the loop derives from the lane index (LDS offsets of the belief phase, stream offsets, ...) is two or three
        // integer instructions away from it; hoisted out of the persistent loop -- which is what the compiler does with loop
        // invariants -- they occupied (and spilled) ~30 vector registers.  An opaque copy per iteration keeps them local.
        int lane = lane_entry;
        asm volatile("" : "+v"(lane));
        lane &= 63;
        int ti = 0;
        if (lane == 0) ti = atomicAdd(&ctl[0], 1);
        ti = __builtin_amdgcn_readfirstlane(ti);
        const bool valid = ti < ntl;
        const int li = valid ? (a.reverse ? ntl - 1 - ti : ti) : 0;      // position in the workgroup's walk
        const int t = STRIDED ? li * (int)gridDim.x + (int)blockIdx.x : tb + li;
        GBP_PH(0);                                         // ticket
        if (!valid) {                                      // no tile left: the landmark beliefs of this wave's last tile, and out
            if (pend && !GBP_DBG(a, 4)) tile_landmark_beliefs(p, wl, lane, q_t, q_l0, q_nl, pre);
            GBP_PH_NOWAIT(2);
            break;
        }

        // everything the factor streams
        // (unconditional: the slots of a tile exist in storage for all 64 lanes, and straight-line loads need no merge
        //  copies that would make the wave wait for them before the tail below)
        // this tile streams past the memory-side cache: FusedArgs::pin; every tile of the general sweep (the cache is left to the staged
        // camera rows: nontemporal message stores as well as loads, 116.6-117.3 against 119.4-119.7 us per sweep with plain stores)
        const bool past = STAGED || (PINNED && li >= a.pin);
        issue_streams<LOSS, STAGED>(p, t, lane, S, past ? 3 : (PINNED ? a.nt : 0));      // (FusedArgs::nt: experiments with the pinned variant only)
#if GBP_PF_DIST > 0
        Touch pf;
        bool pf_on = false;
        if (!STAGED) {
            const int tip = ti + GBP_PF_DIST;                  // (wave-uniform)
            if (tip < ntl) {
                const int tp = tb + (a.reverse ? ntl - 1 - tip : tip);
                if (PINNED && (tp - tb) >= a.pin) touch_tile<true>(p, tp, lane, pf); else touch_tile<false>(p, tp, lane, pf);
                pf_on = true;
            }
        }
#endif
        const unsigned lo = (unsigned)lane * 16u;
        double x0[9], z[2], avar = p.sigma2, qC[2], qL[2], WC[3], VL[3], muC[6], PC[21];
        const unsigned long long words = (unsigned long long)__double_as_longlong(S.a[5].y);      // meta (low) | state (high): gbp_kernels.hpp ROW_SM
        const unsigned meta = (unsigned)words;
        int st = (int)(words >> 32);
        x0[0] = S.a[0].x; x0[1] = S.a[0].y; x0[2] = S.a[1].x; x0[3] = S.a[1].y; x0[4] = S.a[2].x; x0[5] = S.a[2].y; x0[6] = S.a[3].x; x0[7] = S.a[3].y;
        x0[8] = S.a[4].x; z[0] = S.a[4].y; z[1] = S.a[5].x;
        if (LOSS != 0) avar = S.avar;
        qC[0] = S.m[0].x; qC[1] = S.m[0].y; qL[0] = S.m[1].x; qL[1] = S.m[1].y;
        WC[0] = S.m[2].x; WC[1] = S.m[2].y; WC[2] = S.m[3].x; VL[0] = S.m[3].y; VL[1] = S.m[4].x; VL[2] = S.m[4].y;
        // the tile's descriptor, and the heads (mean | covariance | rows) of its landmark records: ten doubles of every twenty,
        // fetched by the whole wave.  (Behind the streams, which need nothing but the tile index: the descriptor's round trip
        // overlaps theirs instead of preceding it.)
        const int4 td = tiles[t];
        const int l0 = td.x, nl = td.y, nf = td.z, maxrank = td.w;
        const bool active = lane < nf;
        const int nhead2 = nl * (LHEAD / 2);               // in 16-byte pieces (a chunk tile: the one landmark it holds a piece of)
        const double2 *lsrc = reinterpret_cast<const double2 *>(p.lrec + (size_t)l0 * LREC);
        constexpr int NSTAGE = (TILE_LMKS * (LHEAD / 2) + 63) / 64;
        double2 stage[NSTAGE];
#pragma unroll
        for (int j = 0; j < NSTAGE; ++j) {
            const int i = j * 64 + lane, rec = (i * 205) >> 10, piece = i - rec * (LHEAD / 2);    // i / 5, exact for i < 128
            stage[j] = i < nhead2 ? lsrc[rec * (LREC / 2) + piece] : make_double2(0.0, 0.0);            // (heads: every other 80 bytes)
        }

        // priors | slot ranges for THIS tile's belief phase, which runs one iteration from now: fetched with the tile's streams (a whole
        // iteration of slack: beyond the memory-side cache they come from HBM, and fetched at the end of the iteration the belief phase
        // of the next one waited for them)
        LmkPre pre_next;
        lmk_prefetch(p, lane, t, l0, nl, pre_next);
        asm volatile("" ::: "memory");
        GBP_PH_NOWAIT(1);                                  // issue of the stream loads

        // ---- tail of the previous tile: its landmark beliefs = prior + messages in adj_factors order (gbp.py:182-193)
        if (pend && !GBP_DBG(a, 4)) tile_landmark_beliefs(p, wl, lane, q_t, q_l0, q_nl, pre);
        asm volatile("" ::: "memory");
        GBP_PH_NOWAIT(2);                                  // landmark beliefs of the previous tile (LDS)

        // the camera record of this tile's factors: a gather that needs `meta` (L2 hits)
        const int cam = active ? (int)(meta >> META_LMK_BITS) : 0;
        GBP_PH(3);                                         // the streams (and meta) have arrived
        {
            const unsigned co = (unsigned)(GBP_DBG(a, 8) ? (cam & 7) : cam) * (unsigned)(CAMREC * 8);      // (dbg 8: what would a cheap gather buy?)
            double v[CAMHEAD];
#pragma unroll
            for (int i = 0; i < CAMHEAD / 2; ++i) { const double2 t2 = ld2(p.cbel, co + 16u * i); v[2 * i] = t2.x; v[2 * i + 1] = t2.y; }
#pragma unroll
            for (int i = 0; i < 6; ++i) muC[i] = v[CAM_MU + i];
#pragma unroll
            for (int i = 0; i < 21; ++i) PC[i] = v[CAM_COV + i];
        }
        asm volatile("" ::: "memory");
        GBP_PH(4);                                         // camera gather
#if GBP_PF_DIST > 0
        if (pf_on) touch_retire(pf);                       // (issued right behind the streams: long since back when the gather has arrived)
#endif

        // landmark heads -> wave scratch -> the lanes of their factors
#pragma unroll
        for (int j = 0; j < NSTAGE; ++j)
            if (j * 64 + lane < TILE_LMKS * (LHEAD / 2)) reinterpret_cast<double2 *>(wl)[j * 64 + lane] = stage[j];
        wave_lds_sync();
        GBP_PH_NOWAIT(5);                                  // landmark records through LDS

        double MCn[21], eC[6];
        int tile_relin = 0;                                // factors of this tile that relinearised (valid in its active lanes)
        // PINNED only.  The compiler guards the first reuse of the gather's registers, at the top of the NEXT iteration, with a wait for
        // everything older than that iteration's own loads: on the path around the `if (active)` block below (no lane active: never the
        // case, a tile has a factor, but it cannot know) the gather counts as still in flight at the loop's back edge.  In the plain
        // kernel that guard is s_waitcnt vmcnt(11) with thirteen loads outstanding, and removing it bought nothing (EXPERIMENTS.md round 6).
        // Here the stream loads sit in two branches each (nontemporal or not, per tile), the guard cannot count them and becomes
        // s_waitcnt vmcnt(0): every wave waited for ALL its streams in front of the belief phase that is there to run while they fly.
        // So: the geometry first (it needs no belief: the gather's round trip), then EVERY lane waits for the gather, outside the block.
        Lin geom;
        if (PINNED) {
            if (active) {
                Params q = p;
                pin_scalars(q);
                factor_geometry(q, x0, z, geom);
            }
            // (all 27 values: which of the gather's loads goes out last is the compiler's choice)
            asm volatile("" ::"v"(muC[0]), "v"(muC[1]), "v"(muC[2]), "v"(muC[3]), "v"(muC[4]), "v"(muC[5]), "v"(PC[0]), "v"(PC[1]), "v"(PC[2]), "v"(PC[3]),
                         "v"(PC[4]), "v"(PC[5]), "v"(PC[6]), "v"(PC[7]));
            asm volatile("" ::"v"(PC[8]), "v"(PC[9]), "v"(PC[10]), "v"(PC[11]), "v"(PC[12]), "v"(PC[13]), "v"(PC[14]), "v"(PC[15]), "v"(PC[16]), "v"(PC[17]),
                         "v"(PC[18]), "v"(PC[19]), "v"(PC[20]));
        }
        if (active) {
            double MLn[6], eL[3], muL[3];
            const double *lhead = wl + (meta & ((1u << META_LMK_BITS) - 1u)) * LHEAD;     // intact until the messages go in below
#pragma unroll
            for (int k = 0; k < 3; ++k) muL[k] = lhead[LR_MU + k];
            Params q = p;
            pin_scalars(q);
            double *lin_w = p.lin + (size_t)t * (LIN_ROWS * WTILE), *msg_w = p.msg + (size_t)t * (MSG_ROWS * WTILE);
            const bool relin = factor_core<LOSS, false>(q, x0, z, st, avar, muC, PC, muL,
                                                        [lhead](double (&c)[6]) {
#pragma unroll
                                                            for (int k = 0; k < 6; ++k) c[k] = lhead[LR_COV + k];
                                                        },
                                                        [lin_w, lo](const double (&x)[9]) {
                                                            st2(lin_w, lo, x[0], x[1]); st2(lin_w, 1024u + lo, x[2], x[3]);
                                                            st2(lin_w, 2048u + lo, x[4], x[5]); st2(lin_w, 3072u + lo, x[6], x[7]);
                                                            st1(lin_w, 4096u + lo, x[8]);
                                                        },
                                                        qC, qL, WC, VL, eC, eL, MCn, MLn, nullptr, PINNED ? &geom : nullptr);
            tile_relin = relin_in_wave(relin);
            n_relin += tile_relin;
            GBP_PH_NOWAIT(6);                              // the maths
            // (one wave-uniform branch around the whole block: five branches, one per store, cost the pinned variant 2-3 us per sweep)
#define GBP_MSG_STORES(ST)                                                                                            \
            ST(msg_w, lo, qC[0], qC[1]); ST(msg_w, 1024u + lo, qL[0], qL[1]);                                         \
            wave_lds_sync();                               /* every lane of the tile has read its landmark head */   \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) wl[lane * 9 + k] = eL[k];                                   \
            ST(msg_w, 2048u + lo, WC[0], WC[1]); ST(msg_w, 3072u + lo, WC[2], VL[0]); ST(msg_w, 4096u + lo, VL[1], VL[2]); \
            _Pragma("unroll") for (int k = 0; k < 6; ++k) wl[lane * 9 + 3 + k] = MLn[k];
            if (past) { GBP_MSG_STORES(st2_nt) } else { GBP_MSG_STORES(st2) }
#undef GBP_MSG_STORES
            if (st != (int)(words >> 32))                   // the state word (high half of ROW_SM): only a factor that did more than age has a new one
                *reinterpret_cast<int *>(reinterpret_cast<char *>(lin_w) + 5120u + lo + 12u) = st;
            if (LOSS != 0) *reinterpret_cast<double *>(reinterpret_cast<char *>(p.avar + (size_t)t * WTILE) + (unsigned)lane * 8u) = avar;
        }
        pre = pre_next;
        if (STAGED) {
            // camera-message rows -> cstage through LDS: a lane-per-factor store would touch 64 different lines per instruction;
            // transposed, eight whole 128-byte lines go out per instruction, 16 bytes per lane
            if (active) {
                double2 *row = reinterpret_cast<double2 *>(wr + lane * CSTAGE_PLAIN);
                row[0] = make_double2(x0[0], x0[1]); row[1] = make_double2(x0[2], x0[3]); row[2] = make_double2(x0[4], x0[5]);
                row[3] = make_double2(x0[6], x0[7]); row[4] = make_double2(x0[8], qC[0]); row[5] = make_double2(qC[1], WC[0]);
                row[6] = make_double2(WC[1], WC[2]); row[7] = make_double2(0.0, 0.0);
                wpos[lane] = S.cpos;
            }
            wave_lds_sync();
            // x0 moves only when a factor relinearises: in every other tile the first 64 bytes of its rows (x0[0..7]) are already what
            // the last sweep staged, and only the second half -- x0[8] | q_C | W | pad -- goes out: 64 instead of 128 bytes per factor
            if (a.full_rows || __builtin_amdgcn_readfirstlane(tile_relin) != 0) {
                const int g8 = lane >> 3, k2 = lane & 7;
                for (int f = g8; f < nf; f += 8)
                    reinterpret_cast<double2 *>(p.cstage + (size_t)wpos[f] * CSTAGE_PLAIN)[k2] = reinterpret_cast<const double2 *>(wr + f * CSTAGE_PLAIN)[k2];
            } else {
                const int g16 = lane >> 2, k2 = 4 + (lane & 3);
                for (int f = g16; f < nf; f += 16)
                    reinterpret_cast<double2 *>(p.cstage + (size_t)wpos[f] * CSTAGE_PLAIN)[k2] = reinterpret_cast<const double2 *>(wr + f * CSTAGE_PLAIN)[k2];
            }
            wave_lds_sync();
            pend = true; q_t = t; q_l0 = l0; q_nl = nl;
            continue;
        }
        wave_lds_sync();
        GBP_PH_NOWAIT(7);                                  // stores issued
        // camera accumulation strictly in tile order
        if (!GBP_DBG(a, 1)) while (__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ti) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
        GBP_PH_NOWAIT(8);                                  // waiting for the accumulation turn
        const int rank = state_rank(st);
        const int cloc = WINDOWED ? (active ? (int)reinterpret_cast<const unsigned short *>(ctl + 2 + win_rows_ints(a.acc_doubles / 27))[cam - cam_base] : 0) : cam - cam_base;
        const bool mine = active && (WINDOWED || (unsigned)cloc < (unsigned)cam_count);
        // Lanes of a tile that hit the same camera add in rank (= lane) order.  Few of them: one round per rank, one lane per camera in
        // every ds_add_f64.  Many (graphs with a few dozen cameras: fr1desk has 63, and up to eight factors of a tile on one of them):
        // ALL lanes in one instruction -- the LDS atomic unit applies the lanes that share an address in ascending lane order, which IS
        // the rank order, so the sums are bitwise those of the rounds (tests/test_edge_shapes_gpu.py pins that on every run) at
        // 27 instead of 27 x (maxrank + 1) instructions: fr1desk_small 15.05 -> 12.35 us per sweep, fr1desk 16.17 -> 13.73.  With one or
        // two duplicates per tile the rounds are faster (1M factors x 500 cameras: 74.3 against 75.2 us per step).
        // The choice is per graph (a kernel variant, fused_plan: by the number of cameras): a per-tile test cost the headline 0.8 us per sweep.
        for (int r = 0; r <= ((GBP_DBG(a, 32) || SINGLE) ? 0 : maxrank); ++r) {      // (dbg 32: first round only -- drops the duplicates, timing experiment)
            if (mine && (SINGLE || rank == r) && !GBP_DBG(a, 2)) {
                double *dst = acc + cloc * 27;
#pragma unroll
                for (int k = 0; k < 6; ++k) unsafeAtomicAdd(dst + k, eC[k]);
#pragma unroll
                for (int k = 0; k < 21; ++k) unsafeAtomicAdd(dst + 6 + k, MCn[k]);
            }
        }
        wave_lds_sync();
        if (lane == 0) __hip_atomic_store(&ctl[1], ti + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pend = true; q_t = t; q_l0 = l0; q_nl = nl;
    }
    if (lane == 0) relin_add(p, n_relin);
    if (STAGED) return;
    GBP_PH_NOWAIT(9);
    __syncthreads();
    GBP_PH_NOWAIT(10);                                     // waiting for the other waves of the workgroup
    // table layout [camera][workgroup][27]: 216-byte runs here, one contiguous 55 KB read per camera in k_cam_reduce_tree
    for (int i = tid; i < (WINDOWED ? cam_count : a.acc_doubles / 27) * (TROW / 2); i += NWAVES * 64) {
        const int c = i / (TROW / 2), k = 2 * (i - c * (TROW / 2));
        const double2 v = make_double2(acc[c * 27 + k], k + 1 < 27 ? acc[c * 27 + k + 1] : 0.0);
        const size_t row = WINDOWED ? (size_t)(ctl + 2)[c] : (size_t)(cam_base + c) * gridDim.x + blockIdx.x;
        *reinterpret_cast<double2 *>(a.block_partials + row * TROW + k) = v;
    }
    GBP_PH(11);                                            // table write-out
    GBP_PH_FLUSH(a.phase, blockIdx.x * NWAVES + wave);
#ifdef GBP_END_STAMP                                        // (tools/boundary_probe.py: when did the LAST workgroup get here?)
    if (a.clk && tid == 0) atomicMax(a.clk + 1, (unsigned long long)wall_clock64());
#endif
}

// One workgroup per camera: partial[c] = sum over the per-workgroup tables in a fixed order (bitwise reproducible).
// The camera's n_blocks x 28 run is contiguous (55 KB at 256 workgroups).  Thread (part, pair) = (tid / 14, tid % 14) adds the 16-byte
// piece `pair` of rows part, part + 73, ... straight from memory -- every load instruction of the block reads one contiguous 16 KB
// chunk, four per thread in flight -- then 9 x 28 threads add every 9th partial sum and 27 threads the last nine.  (Rounds 1-3
// copied the run to LDS first and let 9 x 27 threads walk it: 8.3 us per launch, most of it the copy's round trip and a 28-step
// dependent chain per thread.)  With finish != 0 (single GPU: nothing to exchange) the camera belief is completed in place:
// prior + sum, mean and covariance by seven lanes (VariableNode.update_belief gbp.py:182-193), which saves a dependent launch.
constexpr int RED_THREADS = 1024;
constexpr int RED_PAIRS = TROW / 2;                 // 16-byte pieces per row
constexpr int RED_PARTS = RED_THREADS / RED_PAIRS;  // 73 partial sums per entry ...
constexpr int RED_G = 9;                            // ... then 9, then 1
constexpr int RED_LDS_DOUBLES = (RED_PARTS + RED_G) * TROW + 28;

// returns entry tid (< 27) of the camera's sum in threads 0..26 (0.0 elsewhere); red = RED_LDS_DOUBLES doubles of LDS.  The caller
// synchronises the block before red is used again.  NT = threads of the block: the 73 x 14 (part, pair) items are dealt out over them,
// so the sums -- and their order -- are the same for every block size (the peer-exchange kernel runs 256-thread blocks).
template <int NT>
GBP_DEV double cam_reduce_sum(const double *__restrict__ src, int n_blocks, double *red, int tid)
{
    double *red2 = red + RED_PARTS * TROW;
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    for (int item = tid; item < RED_PARTS * RED_PAIRS; item += NT) {
        const int part = item / RED_PAIRS, pair = item - part * RED_PAIRS;
        double sx = 0.0, sy = 0.0;
        for (int b0 = part; b0 < n_blocks; b0 += 4 * RED_PARTS) {
            double2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int bb = b0 + j * RED_PARTS; v[j] = bb < n_blocks ? s2[(size_t)bb * RED_PAIRS + pair] : make_double2(0.0, 0.0); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { sx += v[j].x; sy += v[j].y; }
        }
        reinterpret_cast<double2 *>(red)[item] = make_double2(sx, sy);
    }
    __syncthreads();
    if (tid < RED_G * TROW) {
        const int g = tid / TROW, k = tid - g * TROW;
        double s = 0.0;
        for (int q = g; q < RED_PARTS; q += RED_G) s += red[q * TROW + k];
        red2[g * TROW + k] = s;
    }
    __syncthreads();
    double s = 0.0;
    if (tid < 27) {
#pragma unroll
        for (int g = 0; g < RED_G; ++g) s += red2[g * TROW + tid];
    }
    return s;
}

// cam_rows (camera windows, fused_plan): camera c's rows are block_partials[cam_rows[c].x .. + cam_rows[c].y) instead of [c n .. c n + n)
__global__ __launch_bounds__(RED_THREADS) void k_cam_reduce_tree(Params p, const double *__restrict__ block_partials, int n_blocks,
                                                                 double *__restrict__ partial, int finish, PeerOut peer, unsigned long long *clk,
                                                                 const int2 *__restrict__ cam_rows)
{
    __shared__ __attribute__((aligned(16))) double sh[RED_LDS_DOUBLES];
    const int c = blockIdx.x, tid = threadIdx.x;
    clk_begin(clk);
    double *tot = sh + (RED_PARTS + RED_G) * TROW;
    const double pri = tid < 27 ? p.cprior[(size_t)c * 27 + tid] : 0.0;      // (asked for ahead of the tables: one round trip less in the tail)
    const int2 rows = cam_rows ? cam_rows[c] : make_int2(c * n_blocks, n_blocks);
    const double s = cam_reduce_sum<RED_THREADS>(block_partials + (size_t)rows.x * TROW, rows.y, sh, tid);
    if (tid < 27) {
        partial[(size_t)c * 27 + tid] = s;
        tot[tid] = s + pri;
    }
    if (peer.n && tid < 64) peer_push_row(peer, c, s, tid);   // sharded, peer-store exchange: straight into every rank's mailbox
    if (!finish) return;
    __syncthreads();
    double *rec = p.cbel + (size_t)c * CAMREC;
    if (tid >= 64 && tid < 64 + 27) p.cbelief[(size_t)c * CBEL + tid - 64] = tot[tid - 64];      // eta | Lambda: one store instruction of another wave
    if (tid < 7) {                                          // mean and the six columns of the covariance, one lane each
        double v[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = tot[k];
        cam_belief_store(v, rec, tid);
    }
#ifdef GBP_END_STAMP
    if (clk && tid == 0) atomicMax(clk + 1, (unsigned long long)wall_clock64());
#endif
}

// Camera windows leave a camera a handful of rows (those of the workgroups whose camera set holds it: two to five in a sequence, where
// k_cam_reduce_tree's 1024 threads per camera took 43.7 us for 10 000 cameras): one WAVE per camera.  Lane (g, pair) = (lane / 14,
// lane % 14), g < 4, adds the 16-byte piece `pair` of rows g, g + 4, ...; the four partial sums are added in the order of g.  Returns
// entry `lane` (< 27) of the sum.  fused_plan picks this form when the cameras have at most ROWS_WAVE_MAX rows on average (fr1desk_small
// with windows forced: 41 rows per camera, 5.1 us in this form against 3.6 in the tree form).
constexpr int ROWS_WAVE_MAX = 16;
constexpr int ROWS_THREADS = 256;
GBP_DEV double cam_rows_sum_wave(const double *__restrict__ src, int n_rows, int lane)
{
    const int g = lane / RED_PAIRS, pair = lane - g * RED_PAIRS;
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double sx = 0.0, sy = 0.0;
    if (g < 4)
        for (int r0 = g; r0 < n_rows; r0 += 32) {           // eight loads in flight per lane, added in row order
            double2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int r = r0 + 4 * j; v[j] = r < n_rows ? s2[(size_t)r * RED_PAIRS + pair] : make_double2(0.0, 0.0); }
#pragma unroll
            for (int j = 0; j < 8; ++j) { sx += v[j].x; sy += v[j].y; }
        }
    double ax = sx, ay = sy;
#pragma unroll
    for (int j = 1; j < 4; ++j) { ax += __shfl(sx, (lane + j * RED_PAIRS) & 63, 64); ay += __shfl(sy, (lane + j * RED_PAIRS) & 63, 64); }      // (meaningful in lanes < 14)
    const double ex = __shfl(ax, lane >> 1, 64), ey = __shfl(ay, lane >> 1, 64);
    return lane < 27 ? ((lane & 1) ? ey : ex) : 0.0;
}

__global__ __launch_bounds__(ROWS_THREADS) void k_cam_reduce_rows(Params p, const double *__restrict__ block_partials, const int2 *__restrict__ cam_rows,
                                                                  double *__restrict__ partial, int finish, PeerOut peer, unsigned long long *clk)
{
    clk_begin(clk);
    const int lane = threadIdx.x & 63, c = blockIdx.x * (ROWS_THREADS / 64) + (threadIdx.x >> 6);
    if (c >= p.C) return;
    const double pri = lane < 27 ? p.cprior[(size_t)c * 27 + lane] : 0.0;
    const int2 rows = cam_rows[c];
    const double s = cam_rows_sum_wave(block_partials + (size_t)rows.x * TROW, rows.y, lane);
    if (lane < 27) partial[(size_t)c * 27 + lane] = s;
    if (peer.n) peer_push_row(peer, c, s, lane);
    if (!finish) return;
    const double tot = s + pri;
    if (lane < 27) p.cbelief[(size_t)c * CBEL + lane] = tot;
    double v[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = __shfl(tot, k, 64);
    cam_belief_store(v, p.cbel + (size_t)c * CAMREC, lane);
}

// Sharded sweep with the peer-store exchange, everything after the sweep kernel in ONE launch: a small grid of persistent
// workgroups first reduces and PUSHES all of its cameras (workgroup tables -> 27 sums -> row c of every rank's mailbox + tag),
// then finishes them, one wave per camera: wait for the n_ranks tags of row c, add the parts in rank order, prior, mean | covariance.
// Every workgroup of every rank pushes before it waits, and the grid is never larger than what is resident at once (256-thread
// workgroups: eight per CU of an MI355X; one camera per workgroup up to 2048 cameras, several beyond; fused_launch caps the grid at
// what the occupancy query admits), so ranks cannot wait for each other in a cycle.  (Round 3 ran 1024-thread workgroups bound to 64
// VGPRs so that two fit a CU; the seven-lane mean | covariance solve of the finish spilled 55 of them: +4 us per sweep.)  Against
// reduce -> finish as two launches this saves a kernel boundary and the global "all rows are out" hand-off; against RCCL also the
// collective's launch and sync.
constexpr int XCHG_THREADS = 256;
constexpr int XCHG_BLOCKS = 2048;
template <bool WAVE_ROWS>                                   // WAVE_ROWS: one wave per camera adds its few rows (camera windows, cam_rows_sum_wave)
__global__ __launch_bounds__(XCHG_THREADS) void k_cam_reduce_xchg(Params p, const double *__restrict__ block_partials, int n_blocks,
                                                                  double *__restrict__ partial, PeerOut peer, PeerWait wait, unsigned long long *clk,
                                                                  const int2 *__restrict__ cam_rows)
{
    __shared__ __attribute__((aligned(16))) double sh[RED_LDS_DOUBLES];
    const int tid = threadIdx.x;
    clk_begin(clk);
    const int wave = tid >> 6, lane = tid & 63;
    if (WAVE_ROWS) {
        for (int c = blockIdx.x * (XCHG_THREADS / 64) + wave; c < p.C; c += (XCHG_THREADS / 64) * gridDim.x) {
            const int2 rows = cam_rows[c];
            const double s = cam_rows_sum_wave(block_partials + (size_t)rows.x * TROW, rows.y, lane);      // the same order as k_cam_reduce_rows
            if (lane < 27) partial[(size_t)c * 27 + lane] = s;
            peer_push_row(peer, c, s, lane);
        }
    } else {
        for (int c = blockIdx.x; c < p.C; c += gridDim.x) {
            const int2 rows = cam_rows ? cam_rows[c] : make_int2(c * n_blocks, n_blocks);
            const double s = cam_reduce_sum<XCHG_THREADS>(block_partials + (size_t)rows.x * TROW, rows.y, sh, tid);      // the same order as k_cam_reduce_tree: bitwise the same sums
            if (tid < 64) {
                if (tid < 27) partial[(size_t)c * 27 + tid] = s;
                peer_push_row(peer, c, s, tid);
            }
            __syncthreads();                                 // sh is overwritten by the next camera
        }
    }
    for (int c = blockIdx.x + wave * gridDim.x; c < p.C; c += (XCHG_THREADS / 64) * gridDim.x) cam_finish_wave(p, nullptr, peer.n, 0, wait, c, lane);
#ifdef GBP_END_STAMP
    __syncthreads();
    if (clk && tid == 0) atomicMax(clk + 1, (unsigned long long)wall_clock64());
#endif
}


// ---- the SINGLE accumulation's hardware assumption, checked where it is relied on ---------------------------------------------
// SINGLE (k_sweep_wat<.., SINGLE = true>) lets every lane of a tile issue its ds_add_f64 in ONE instruction even when several lanes hit
// the same table entry, and counts on the LDS atomic unit applying same-address lanes in ASCENDING LANE ORDER -- the order of the
// rank rounds -- for sums that are bitwise those of the rounds variant (equal across ranks, run to run, and across the checkpoint
// tests).  No ISA document promises that order; it is what gfx950 does (EXPERIMENTS.md round 4).  So the plan of every graph that would
// select SINGLE first runs this probe on the device it will run on (once per device and process): one wave, seven address patterns
// from "all 64 lanes on one entry" to "eight lanes each on eight entries", addends of wildly different magnitudes (so that any other
// order rounds differently), compared bitwise with the same additions made one by one in lane order.  On a mismatch the plan falls
// back to the rounds variant and says so (gbp_ba_plan_info).
GBP_DEV int single_probe_addr(int lane, int pat)
{
    switch (pat) {
    case 0: return 0;
    case 1: return lane & 1;
    case 2: return lane % 3;
    case 3: return lane >> 3;
    case 4: return (lane * 7) & 7;
    case 5: return lane < 40 ? 0 : (lane & 7);
    default: return (lane * lane + 3 * lane) & 7;
    }
}
constexpr int SINGLE_PROBE_PATTERNS = 7;
__global__ __launch_bounds__(64) void k_single_probe(int *out)
{
    __shared__ double acc[8], val[64];
    const int lane = threadIdx.x;
    int bad = 0;
    for (int pat = 0; pat < SINGLE_PROBE_PATTERNS; ++pat) {
        const unsigned hsh = ((unsigned)lane * 2654435761u + (unsigned)pat * 40503u) >> 7;
        val[lane] = ldexp((double)(hsh & 0xfffffu) + 0.5, (int)((lane * 13 + pat * 7) % 41) - 20) * ((hsh >> 21) & 1u ? -1.0 : 1.0);
        if (lane < 8) acc[lane] = 0.1 * (lane + 1);
        __syncthreads();
        unsafeAtomicAdd(&acc[single_probe_addr(lane, pat)], val[lane]);      // the sweep's instruction: ds_add_f64, all lanes at once
        __syncthreads();
        if (lane < 8) {
            double e = 0.1 * (lane + 1);
            for (int l = 0; l < 64; ++l)
                if (single_probe_addr(l, pat) == lane) e += val[l];          // (val[] is already rounded: nothing to contract)
            if (__double_as_longlong(e) != __double_as_longlong(acc[lane])) bad |= 1 << pat;
        }
        __syncthreads();
    }
    if (bad) atomicOr(out, bad);
}

// 1: the device adds same-address lanes in lane order, 0: it does not (mask of failing patterns in *mask), < 0: the probe could not run
inline int single_probe(hipStream_t stream, int *mask)
{
    static std::mutex mtx;
    static std::vector<int> cache;                          // per device: -1 unknown, else the failing-pattern mask
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lock(mtx);
    if ((int)cache.size() <= dev) cache.resize(dev + 1, -1);
    if (cache[dev] < 0) {
        int *d = nullptr, v = 0;
        if (hipMalloc(reinterpret_cast<void **>(&d), sizeof(int)) != hipSuccess) return -1;
        bool ok = hipMemsetAsync(d, 0, sizeof(int), stream) == hipSuccess;
        if (ok) { hipLaunchKernelGGL(k_single_probe, dim3(1), dim3(64), 0, stream, d); ok = hipGetLastError() == hipSuccess; }
        ok = ok && hipMemcpyAsync(&v, d, sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
        (void)hipFree(d);
        if (!ok) return -1;
        cache[dev] = v;
    }
    int v = cache[dev];
    if (const char *e = getenv("GBP_SINGLE_PROBE_FAIL")) v = atoi(e);          // test switch: pretend the probe saw this mask
    if (mask) *mask = v;
    return v == 0 ? 1 : 0;
}

// ------------------------------------------------------------------------------------ host --

// (Rounds 2-3 added the messages to a second group of up to 758 cameras in an extra pass over the stored messages, k_cam_pass: 138.7 us
//  per sweep at C = 1000.  The general sweep's persistent STAGED form does the same graph in 127 us and has no camera limit: removed.)

// Workgroup tile ranges + per-workgroup camera tables.
// The fused sweep's variants -- loss x PINNED x SINGLE x WINDOWED (each a compile-time property of the persistent loop: a run-time
// test per tile cost the headline 0.8-1.5 us per sweep) -- as one table, for the plan (LDS attribute) and the launch.
using SweepKernel = void (*)(Params, FusedArgs, const int4 *);
template <bool PINNED, bool SINGLE, bool WINDOWED>
inline SweepKernel sweep_variant_of_loss(int loss)
{
    switch (loss) {
    case 0: return k_sweep_wat<0, WAT_WAVES, false, PINNED, SINGLE, WINDOWED>;
    case 1: return k_sweep_wat<1, WAT_WAVES, false, PINNED, SINGLE, WINDOWED>;
    default: return k_sweep_wat<2, WAT_WAVES, false, PINNED, SINGLE, WINDOWED>;
    }
}
inline SweepKernel sweep_variant(int loss, bool pinned, bool single, bool windowed)
{
    switch ((pinned ? 1 : 0) + (single ? 2 : 0) + (windowed ? 4 : 0)) {
    case 0: return sweep_variant_of_loss<false, false, false>(loss);
    case 1: return sweep_variant_of_loss<true, false, false>(loss);
    case 2: return sweep_variant_of_loss<false, true, false>(loss);
    case 3: return sweep_variant_of_loss<true, true, false>(loss);
    case 4: return sweep_variant_of_loss<false, false, true>(loss);
    case 5: return sweep_variant_of_loss<true, false, true>(loss);
    case 6: return sweep_variant_of_loss<false, true, true>(loss);
    default: return sweep_variant_of_loss<true, true, true>(loss);
    }
}

// wg_win / wg_cams (n_win = workgroups, or 0): the workgroups' camera windows -- build_graph's decision (k_wg_cam_sets: per workgroup
// {lowest camera, cameras in its set, offset into wg_cams, width of its interval}); without them every workgroup's table covers all cameras.
inline int fused_plan(FusedPlan &pl, const Params &p, hipStream_t stream, int n_cus, const int4 *wg_win = nullptr, const int *wg_cams = nullptr, int n_win = 0)
{
    if (p.F == 0 || p.C == 0 || p.T == 0) return 0;
    pl.n_blocks = std::max(1, std::min(p.T, n_cus));
    if (const char *nb = getenv("GBP_FUSED_BLOCKS")) pl.n_blocks = std::max(1, std::min(pl.n_blocks, atoi(nb)));   // experiment switch
    pl.windowed = (wg_win && wg_cams && n_win == pl.n_blocks) ? 1 : 0;
    pl.rows_wave = 0;
    // the sweep's table shares the LDS with the waves' scratch: more cameras than fit run the general sweep (STAGED form of the same loop)
    const int cmax = fused_max_cams();
    pl.max_window = pl.max_width = 0;
    size_t table_rows = (size_t)pl.n_blocks * p.C;
    std::vector<int2> cam_rows;
    std::vector<int> rowidx;
    if (pl.windowed) {
        // The rows of block_partials stay CAMERA-major -- the reduce reads one contiguous run per camera -- but a camera has rows only
        // for the workgroups whose set holds it, in workgroup order.
        cam_rows.assign((size_t)p.C, make_int2(0, 0));
        table_rows = 0;
        for (int b = 0; b < pl.n_blocks; ++b) {
            const int4 w = wg_win[b];
            if (w.y < 0 || w.z != (int)table_rows || (w.y && (w.x < 0 || w.x + w.w > p.C))) return -1;
            for (int k = 0; k < w.y; ++k) {
                const int c = wg_cams[(size_t)w.z + k];
                if (c < w.x || c >= w.x + w.w) return -1;
                cam_rows[(size_t)c].y++;
            }
            table_rows += (size_t)w.y;
            pl.max_window = std::max(pl.max_window, w.y); pl.max_width = std::max(pl.max_width, w.w);
        }
        if (table_rows > (size_t)INT32_MAX || pl.max_width > 65536) return -1;
        int first = 0;
        for (int c = 0; c < p.C; ++c) { cam_rows[(size_t)c].x = first; first += cam_rows[(size_t)c].y; cam_rows[(size_t)c].y = 0; }
        rowidx.resize(std::max<size_t>(table_rows, 1));
        for (int b = 0; b < pl.n_blocks; ++b)
            for (int k = 0; k < wg_win[b].y; ++k) {
                int2 &cr = cam_rows[(size_t)wg_cams[(size_t)wg_win[b].z + k]];
                rowidx[(size_t)wg_win[b].z + k] = cr.x + cr.y++;
            }
        // few rows per camera ON AVERAGE: one wave adds them (k_cam_reduce_rows).  A wave takes 32 rows per round trip, so one camera
        // with many rows costs that launch microseconds where the tree form costs every camera a 1024-thread workgroup -- which is also
        // why MANY cameras take the wave form whatever their rows (tools/manycam_probe.sh, random cameras, us per launch tree / wave:
        // 2 000 cameras x 29 rows 11.4 / 6.8, x 56 rows 12.1 / 8.1; 5 000 x 24 24.8 / 9.6; 1 000 x 99 8.3 / 7.7; 500 x 100 5.4 / 6.5, x 160
        // 5.9 / 7.7: tree ~ 2.6 + 0.004 C + 0.017 R, wave ~ 5.2 + 0.0003 C + 0.025 R with R in thousands of rows).
        const char *e = getenv("GBP_ROWS_WAVE_MAX");        // (tests: 0 keeps the tree form)
        const int wave_max = e ? atoi(e) : ROWS_WAVE_MAX;
        pl.rows_wave = (table_rows <= (size_t)wave_max * (size_t)p.C || (!e && (double)p.C > 662.0 + 2.16e-3 * (double)table_rows)) ? 1 : 0;
        if (fused_shmem_windows(pl.max_window, pl.max_width) > (size_t)LDS_BYTES) { pl.windowed = 0; return 0; }
    }
    if (!pl.windowed && p.C > cmax) return 0;
    pl.group_cams = pl.windowed ? std::max(pl.max_window, 1) : p.C;
    pl.n_groups = 1;
    const int acc_doubles = pl.group_cams * 27;
    const size_t shmem = pl.windowed ? fused_shmem_windows(pl.group_cams, pl.max_width) : fused_shmem(pl.group_cams);
    double *d_bp = nullptr;                                 // (workgroup b walks tiles [b T / n_blocks, (b + 1) T / n_blocks): computed in the kernels)
    pl.table_rows = (long long)table_rows;
    if (fused_upload<double>(pl, &d_bp, nullptr, table_rows * TROW, stream)) return -1;
    pl.args = FusedArgs{};
    pl.args.block_partials = d_bp; pl.args.acc_doubles = acc_doubles; pl.args.cam_base = 0; pl.args.cam_count = std::min(p.C, pl.group_cams);
    pl.d_cam_rows = nullptr;
    if (pl.windowed) {
        int4 *d_win = nullptr; int *d_rowidx = nullptr, *d_wgcams = nullptr; int2 *d_cam_rows = nullptr;
        if (fused_upload<int4>(pl, &d_win, wg_win, (size_t)n_win, stream) || fused_upload<int>(pl, &d_rowidx, rowidx.data(), rowidx.size(), stream) ||
            fused_upload<int>(pl, &d_wgcams, wg_cams, table_rows, stream) || fused_upload<int2>(pl, &d_cam_rows, cam_rows.data(), cam_rows.size(), stream)) return -1;
        pl.args.win = d_win; pl.args.rowidx = d_rowidx; pl.args.wgcams = d_wgcams; pl.d_cam_rows = d_cam_rows;
    }
#if defined(GBP_FUSED_DBG_SWITCHES) || defined(GBP_PHASE_TIMING)
    if (instrument_plan(pl, stream)) return -1;             // experimental/gbp_instrument.hpp: GBP_FUSED_DBG, the phase buffer
#endif
    {
        // What a sweep touches, against the 256 MiB memory-side cache.  Everything fits: nothing bypasses it.  Beyond it the first
        // tiles of every workgroup's range -- 160 MiB worth, tables and records included -- keep using the cache and stay resident
        // from sweep to sweep; the remaining tiles stream PAST it, loads and message stores (nontemporal), instead of everything
        // thrashing: 72.8 against 78.0 ps per factor at 1.35M factors, 72.8 / 75.8 at 2M, 68.5 / 69.2 at 10M against the round's
        // earlier policy (lin rows of ALL tiles past the cache, bit 0 of nt, which had brought 1.2M-1.5M from 83 to 76-78), and
        // WORSE below the cache size (71.3 against 67.3 at 1.1M): profiles/r04_size_sweep.jsonl, EXPERIMENTS.md.  The split is
        // per workgroup so that all of them finish together.
        const double S = (double)p.T * WTILE, MiB = 1024.0 * 1024.0;
        const double fixed = (double)table_rows * TROW * 8 + (double)p.C * (CAMREC + CBEL + 27) * 8;
        const double touched = S * (LIN_ROWS + MSG_ROWS) * 8 + S * 8 + (double)p.L * LREC * 8 + fixed;
        const double per_tile = WTILE * (LIN_ROWS + MSG_ROWS) * 8.0 + (double)p.L * LREC * 8 / std::max(p.T, 1);
        // (the share that pays shrinks with the distance from the cache size: 200 MiB just beyond it -- 67.1 against 70.7 ps per factor at
        //  1.15M factors with 160 -- 140 from 1.5M factors on: 73.4 against 76.2 / 78.3 with 180 / 220)
        // (round 6, with the strided walk of the pinned variant -- the cacheable tiles are one contiguous piece of the graph then -- a larger
        //  share pays than the 140-200 MiB above: 1.15M factors 73.6 us per launch with 240 MiB against 76.9 with 200; 1.35M 88.1 with 220,
        //  89.0 with 240, 90.9 with 200, 93.0 with 140; 2M 134.4 with 200, 137.0 with 220, 139.0 with 140, 141.3 with 240; 3M 205.4 with
        //  200, 210.5 with 220; 10M flat: profiles/r06_keep_sweep.txt)
        double keep_mib = touched > 256.0 * MiB ? (touched < 350.0 * MiB ? 230.0 : 200.0) : -1.0;      // < 0: everything stays cacheable
        if (const char *e = getenv("GBP_FUSED_PIN_MIB")) keep_mib = atof(e);
        pl.args.nt = 0;
        pl.args.pin = 0x7fffffff;
        if (keep_mib >= 0.0) pl.args.pin = (int)(std::max(0.0, keep_mib * MiB - fixed) / per_tile / pl.n_blocks);
        // few cameras: many factors of a 60-factor tile share one (fr1desk: 63 cameras, up to eight) -- the SINGLE variant of the accumulation
        pl.single = getenv("GBP_ACC_SINGLE") ? atoi(getenv("GBP_ACC_SINGLE")) : (pl.group_cams <= (pl.args.pin != 0x7fffffff ? 200 : 350) ? 1 : 0);      // (1M factors: 66.1 against 75.1 us per step at 64 cameras, 68.0 / 72.4 at 128, 69.7 / 71.6 at 200, 72.7 / 73.4 at 300, equal at 400, 75.2 / 74.3 at 500; beyond the cache size -- the pinned variant -- 2M factors: 135.4 / 148.3 at 100 cameras, 151.0 / 143.7 at 300)
        if (pl.single) {                                    // the order SINGLE relies on is verified on this device before it is used (single_probe)
            pl.single_probe = single_probe(stream, &pl.single_probe_mask);
            if (pl.single_probe < 0) return -1;
            if (pl.single_probe == 0) {
                pl.single = 0;
                fprintf(stderr, "[gbp] this device does not apply same-address LDS atomics of one instruction in lane order (probe mask 0x%x): "
                                "the fused sweep uses one accumulation round per rank instead\n", pl.single_probe_mask);
            }
        }
        if (const char *e = getenv("GBP_FUSED_NT")) pl.args.nt = atoi(e);      // experiments: bit 0 lin rows, bit 1 message rows of the cacheable tiles
        if (getenv("GBP_PLAN_DEBUG")) fprintf(stderr, "[gbp] fused plan: T %d blocks %d touched %.1f MiB keep %.1f MiB pin %d tiles per workgroup\n", p.T, pl.n_blocks, touched / MiB, keep_mib, pl.args.pin);
    }
    pl.shmem = shmem;
    // (the attribute belongs to the FUNCTION, not to this plan: every handle asks for the whole LDS, so that a later handle with a smaller
    //  table cannot lower what an earlier one launches with)
    for (int v = 0; v < 24; ++v)
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sweep_variant(v % 3, (v / 3) & 1, (v / 6) & 1, (v / 12) & 1)),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess) return -1;
    pl.enabled = true;
    return 0;
}

// returns 0 or a hipError_t value
inline int fused_launch(FusedPlan &pl, const Params &p0, int robustify, int local_relin, double *partial, hipStream_t stream,
                        int finish, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, bool defer_big = false, int reverse = 0,
                        const PeerOut *peer = nullptr, unsigned long long *clk = nullptr, const PeerWait *merged = nullptr)
{
    pl.args.reverse = reverse;
    pl.args.clk = clk;                                      // [0..1] the sweep kernel's stamps, [2..3] the reduce kernel's
    Params p = p0;
    p.robustify = robustify; p.local_relin = local_relin;
    const dim3 grid(pl.n_blocks), block(WAT_WAVES * 64);
    if (e0) (void)hipEventRecord(e0, stream);
    const bool pinned = pl.args.pin != 0x7fffffff;
    hipLaunchKernelGGL(sweep_variant(p.loss, pinned, pl.single != 0, pl.windowed != 0), grid, block, pl.shmem, stream, p, pl.args, p.tiles);
    if (e1) (void)hipEventRecord(e1, stream);
    if (p.parts && !defer_big) hipLaunchKernelGGL(k_lmk_finish_parts, dim3((p.T + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, p);
    const size_t red_shmem = 0;                             // (static LDS)
    PeerOut po{};
    if (peer) po = *peer;
    if (merged && peer) {                                   // reduce -> push -> wait -> finish in one launch (peer-store exchange)
        // The grid must be resident at once (its workgroups wait for other ranks' workgroups of the same index, and the dispatch order
        // is nobody's contract): never more workgroups than the occupancy query admits on this device.
        int xb = pl.xchg_blocks;
        if (xb == 0) {
            int per_cu = 0, dev = 0, cus = 0;
            const void *fn = pl.rows_wave ? reinterpret_cast<const void *>(&k_cam_reduce_xchg<true>) : reinterpret_cast<const void *>(&k_cam_reduce_xchg<false>);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, XCHG_THREADS, red_shmem) != hipSuccess ||
                hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1)
                return (int)hipErrorUnknown;
            xb = std::min(XCHG_BLOCKS, per_cu * cus);
            if (const char *e = getenv("GBP_XCHG_BLOCKS")) xb = std::max(1, std::min(xb, atoi(e)));
            pl.xchg_blocks = xb;
        }
        if (pl.rows_wave)
            hipLaunchKernelGGL(k_cam_reduce_xchg<true>, dim3(std::min((p.C + XCHG_THREADS / 64 - 1) / (XCHG_THREADS / 64), xb)), dim3(XCHG_THREADS), red_shmem, stream, p,
                               pl.args.block_partials, pl.n_blocks, partial, po, *merged, clk ? clk + 2 : nullptr, pl.d_cam_rows);
        else
            hipLaunchKernelGGL(k_cam_reduce_xchg<false>, dim3(std::min(p.C, xb)), dim3(XCHG_THREADS), red_shmem, stream, p, pl.args.block_partials, pl.n_blocks, partial,
                               po, *merged, clk ? clk + 2 : nullptr, pl.d_cam_rows);
        return (int)hipGetLastError();
    }
    if (pl.rows_wave) {
        hipLaunchKernelGGL(k_cam_reduce_rows, dim3((p.C + ROWS_THREADS / 64 - 1) / (ROWS_THREADS / 64)), dim3(ROWS_THREADS), 0, stream, p, pl.args.block_partials,
                           pl.d_cam_rows, partial, finish, po, clk ? clk + 2 : nullptr);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(k_cam_reduce_tree, dim3(p.C), dim3(RED_THREADS), red_shmem, stream, p, pl.args.block_partials, pl.n_blocks, partial, finish, po, clk ? clk + 2 : nullptr, pl.d_cam_rows);
    return (int)hipGetLastError();
}

// the general sweep's factor kernel: the persistent loop in its STAGED form (no table: any number of cameras)
// attr_set: the caller's per-handle flag -- the dynamic-LDS attribute is a property of (kernel, device), and handles of one process may
// sit on different devices (ranks as threads)
inline int staged_launch(const Params &p0, int robustify, int local_relin, int n_cus, int reverse, hipStream_t stream, unsigned long long *clk, int full_rows,
                         bool *attr_set)
{
    Params p = p0;
    p.robustify = robustify; p.local_relin = local_relin;
    FusedArgs a{};
    a.reverse = reverse; a.clk = clk; a.full_rows = full_rows; a.pin = 0x7fffffff;
    const int n_blocks = std::max(1, std::min(p.T, n_cus));
    const size_t shmem = sizeof(double) * ((size_t)WAT_WAVES * STAGED_WAVE_DOUBLES + 1);
    if (!*attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_wat<0, WAT_WAVES, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_wat<1, WAT_WAVES, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_wat<2, WAT_WAVES, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
            return (int)hipErrorUnknown;
        *attr_set = true;
    }
    const dim3 grid(n_blocks), block(WAT_WAVES * 64);
    switch (p.loss) {
    case 0: hipLaunchKernelGGL((k_sweep_wat<0, WAT_WAVES, true>), grid, block, shmem, stream, p, a, p.tiles); break;
    case 1: hipLaunchKernelGGL((k_sweep_wat<1, WAT_WAVES, true>), grid, block, shmem, stream, p, a, p.tiles); break;
    default: hipLaunchKernelGGL((k_sweep_wat<2, WAT_WAVES, true>), grid, block, shmem, stream, p, a, p.tiles); break;
    }
    return (int)hipGetLastError();
}

}  // namespace gbp
