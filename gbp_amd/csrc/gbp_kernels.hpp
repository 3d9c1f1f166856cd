// gbp_kernels.hpp -- data layout + general-purpose HIP kernels of the GBP bundle-adjustment sweep (gfx950).
//
// Data layout in HBM (fp64 unless noted; "packed" = upper triangle, row-major: 6x6 -> 21, 3x3 -> 6)
// -------------------------------------------------------------------------------------------------
// Factors live in TILES of 64 slots (one wavefront).  Factors are ordered landmark-major (stable by
// reference factor id inside a landmark) and a tile owns up to 24 whole landmarks: all factors of those
// landmarks, nf <= 64 of the 64 slots used.  A landmark may also SPAN tiles -- one with more than 64 factors always does ("chunk"
// tiles: nl = 1, a piece of that landmark and nothing else), and in the dense packing (graphs whose whole landmarks would leave the
// tiles under-filled: gbp_capi.hip build_graph) tile t simply holds factors [64 t, 64 t + 64) of the landmark-major list.  A tile that
// holds only a PART of a landmark adds its part of the messages up and writes the sum to Params::parts (row 2 t: the landmark began
// in an earlier tile, row 2 t + 1: it goes on in the next); k_lmk_finish_parts forms those beliefs after the sweep.
// Slot id = tile*64 + lane.
//
// Everything a factor streams per sweep is TILE-CONTIGUOUS; inside a tile rows are stored in pairs,
// [row/2][64 lanes][2], so a lane owns 16 contiguous bytes per pair (one dwordx4 access):
//     lin[tile][12][64]   rows 0-8 x0 = linearisation point (t, w, y)      Factor.linpoint      gbp.py:231
//                         rows 9-10 z  = measurement                        Factor.measurement   gbp.py:233
//                         row  11  two 32-bit words: meta | state (below)   -- they arrive with z[1] in one 16-byte load
//     avar[slot]          adaptive noise variance, robust losses only       gbp.py:242
//     msg[tile][10][64]   rows 0-1 / 2-3 the coefficients q_C / q_L of the two message etas (eta = J^T q with J at x0),
//                         rows 4-6 / 7-9 the 2x2 cores W / V of the two message precisions (Lambda = J^T Q J:
//                         gbp_math.hpp rank2_update)                                  Factor.messages  gbp.py:222
//     state word  int32 = iters_since_relin << 12 | pending << 11 | rank << 2 | robust << 1 | damped     gbp.py:245-249
//     meta word  uint32 = camera index << 8 | landmark's position inside its tile
// so a wave's access is one contiguous 1 KB line per row pair and a tile's whole working set is one
// 5 KB + 6 KB block (the reference's dense messages would be 27 KB: eta 6 + 3, Lambda 6x6 + 3x3 per factor).  (Measured on MI355X, tools/membench.hip: the same bytes streamed as 83 separate
// stride-F arrays reach 3.9 TB/s at 4 waves/CU, as tile-contiguous blocks 5.4 TB/s.)
// Lambda_f / eta_f (90 doubles per factor in the reference) are never stored: they are rebuilt from x0 and z
// every sweep (2x9 Jacobian ~ 150 flops versus 720 bytes of traffic).
//
// A factor reads a belief as MEAN | COVARIANCE (gbp_math.hpp: covariance form).
// Landmarks: array of records lrec[L][20] = mu 3 | Sigma 6 | {first slot, end slot} as two int32 | prior (eta 3 | Lambda 6) | pad.
//            A tile's landmarks are consecutive records.  Their eta | Lambda (a view: VariableNode.belief) is formed from mean |
//            covariance when it is asked for (k_lmk_belief_view): 72 bytes per landmark and sweep that no kernel reads.
// Cameras:   cbel[C][32] = mu 6 | Sigma 21 | pad (224 bytes gathered per factor, L2-resident: 500 cams = 128 KB),
//            cbelief[C][28] = eta 6 | Lambda 21 | pad (views), cprior[C][27] = eta 6 | Lambda 21;  cptr[C+1], cadj[F] = slots of
//            each camera's factors (reference order).
//
// (This header holds the layout and every device-side helper; the kernels themselves live with the translation unit that launches them:
// gbp_sweep_kernels.hpp + gbp_fused.hpp -> gbp_capi_sweep.hip, gbp_view_kernels.hpp -> gbp_capi_views.hip, gbp_build.hpp -> gbp_capi.hip.)
// General sweep (any shape) = k_factor_tile (one wave per tile: messages, the tile's landmark beliefs, camera messages
// staged in camera-major order) -> k_lmk_belief_list (landmarks larger than a tile) -> k_cam_partial_staged (one
// workgroup per camera, contiguous run) -> k_cam_finish.  k_lmk_belief / k_cam_partial form beliefs from the STORED
// messages (update_all_beliefs).  The fused sweep is in gbp_fused.hpp.
#pragma once
#include "gbp_math.hpp"

namespace gbp {

constexpr int WTILE = 64;         // slots per tile = lanes per wavefront
constexpr int TILE_LMKS = 24;     // most landmarks a tile owns
constexpr int LIN_ROWS = 12, MSG_ROWS = 10;
constexpr int ROW_X0 = 0, ROW_Z = 9, ROW_SM = 11;     // ROW_SM: 8 bytes = meta (low word) | state (high word), beside z[1] in the last pair
constexpr int ROW_QC = 0, ROW_QL = 2, ROW_WC = 4, ROW_VL = 7;
constexpr int XTRA_ROW = 9;       // doubles per slot of the dense remainder (num_undamped_iters = 0 only): camera 6 | landmark 3
constexpr int LREC = 20;          // doubles per landmark record: mu 3 | Sigma 6 | rows | prior 9 | pad
constexpr int LR_MU = 0, LR_COV = 3, LR_ROWS = 9, LR_PRIOR = 10;
constexpr int LHEAD = 10;         // ... of which a factor reads the first ten (mean | covariance | rows)
constexpr int CAMREC = 32;        // doubles per camera record: mu 6 | Sigma 21 | pad (256 bytes: a record is exactly two 128-byte lines)
constexpr int CAM_MU = 0, CAM_COV = 6;
constexpr int CAMHEAD = 28;       // ... of which a factor gathers 28 (mean | covariance): 14 x 16 bytes
constexpr int CBEL = 28;          // doubles per row of cbelief: eta 6 | Lambda 21 | pad (views, checkpoints; no sweep kernel reads it)
constexpr int META_LMK_BITS = 8;   // meta = camera << 8 | landmark slot.  (camera in the LOW bits + '& 0xffffff' was
                                   // miscompiled by hipcc 7.2: the mask vanished in front of a v_mad_u64_u32 address multiply)
constexpr int BLOCK = 256;
constexpr int CSTAGE_ROW = 20;     // doubles per row of the camera-major staging buffer: x0 9 | q_C 2 | W 3 | (xtra: remainder 6)
constexpr int CSTAGE_USED = 14;    // ... of which the rows carry 14 unless Params::xtra is set
constexpr int CSTAGE_PLAIN = 16;   // row stride without xtra: one whole, aligned 128-byte line per factor, written in full (a 112-byte row at a
                                   // 160-byte stride straddled two lines and left both partly written: read-modify-write at the memory side)

struct Params {
    int F, T, L, C;               // factors, tiles (slots = 64 T), landmarks, cameras
    Intrinsics K;
    double sigma2, nstds, beta, eta_damping;
    int num_undamped, min_linear, loss;
    int robustify, local_relin;
    double *lin, *msg;            // tile-blocked factor data (lin's last row carries the meta and state words: slot_words)
    double *avar;                 // [slot] adaptive noise variance (gbp.py:242), or NULL when the loss is None (it is sigma^2 then)
    const int4 *tiles;            // {first landmark, landmarks owned, slots used, max rank}
    double *lrec;
    double *cbel, *cprior, *cbelief;
    const int *cptr, *cadj;
    double *cstage;               // general sweep: [F][crow] what rebuilds the camera messages, in camera-major (reference) order, or NULL
    int crow;                     // doubles per staged row: CSTAGE_PLAIN, or CSTAGE_ROW with xtra
    const int *cpos;              // slot -> row of cstage
    double *xtra;                 // [slot][9] out-of-span remainder of the message etas, or NULL (gbp_math.hpp header: only when
                                  // num_undamped_iters = 0 lets a factor be damped in the sweep it relinearises in)
    int reverse_walk;             // general sweep: tiles and cameras are visited backwards (every other sweep; results do not depend on it)
    int *relin_slot;              // this sweep's "factors that relinearised" counter (ba.py:96-99 without a read-back of F words), or NULL
    int stage;                    // 0: a whole synchronous_iteration.  STAGE_* bits: the reference's stage-wise entry points (gbp.py:46-84)
    int clk, clk_inc;             // the graph's relinearisation clock AFTER this call, and whether this call advances it (state word below)
    double *parts;                // [2 T][PART_ROW] partial sums (eta 3 | Lambda 6, no prior) of the landmarks that span more than one tile, or
                                  // NULL when no landmark does (tile_landmark_beliefs writes them, k_lmk_finish_parts adds them up)
};
constexpr int PART_ROW = 10;      // doubles per row of Params::parts (9 + pad: rows start on 16 bytes)
constexpr int STAGE_NO_TEST = 1;      // compute_all_messages alone: no relinearisation test, no iters_since_relin bookkeeping (gbp.py:46-54)
constexpr int STAGE_NO_BELIEFS = 2;   // ... and no belief is touched (the reference updates them in update_all_beliefs only, gbp.py:56-58)

// The sweep's scalar parameters, pinned to SGPRs at the top of every tile: left alone, the compiler copies the ten doubles
// into VGPR pairs once outside the persistent loop (a VALU instruction takes one scalar operand) and then keeps 20 registers
// busy -- or spills them -- through the whole tile.  After this, each use moves its operand in where it is needed.
GBP_DEV void pin_scalars(Params &q)
{
    asm volatile("" : "+s"(q.K.fx), "+s"(q.K.fy), "+s"(q.K.cx), "+s"(q.K.cy));
    asm volatile("" : "+s"(q.sigma2), "+s"(q.nstds), "+s"(q.beta), "+s"(q.eta_damping));
}

// element (slot, row) of a tile block: rows are stored in PAIRS, [row/2][lane][row%2], so that a lane owns 16 contiguous
// bytes per pair and the fused sweep moves a tile with half the vector-memory instructions (a wave can have at most
// 64 of them outstanding; 8-byte rows needed ~130 per tile)
GBP_DEV size_t lin_at(int slot, int row) { return (((size_t)(slot >> 6) * (LIN_ROWS / 2) + (row >> 1)) * WTILE + (slot & 63)) * 2 + (row & 1); }
GBP_DEV size_t msg_at(int slot, int row) { return (((size_t)(slot >> 6) * (MSG_ROWS / 2) + (row >> 1)) * WTILE + (slot & 63)) * 2 + (row & 1); }

// The meta and state words of a slot share the double ROW_SM of the lin block (rounds 1-3 kept them in arrays of their own: 8 bytes per
// slot and sweep more traffic, 8.5 MB more working set at 1M factors, two more loads per tile).
GBP_DEV unsigned *slot_words(const Params &p, int slot) { return reinterpret_cast<unsigned *>(p.lin + lin_at(slot, ROW_SM)); }
GBP_DEV unsigned slot_meta(const Params &p, int slot) { return slot_words(p, slot)[0]; }
GBP_DEV int slot_state(const Params &p, int slot) { return (int)slot_words(p, slot)[1]; }
GBP_DEV void set_slot_state(const Params &p, int slot, int st) { slot_words(p, slot)[1] = (unsigned)st; }
GBP_DEV double slot_avar(const Params &p, int slot) { return p.avar ? p.avar[slot] : p.sigma2; }

// state word: relinearisation clock << 12 | pending << 11 | rank << 2 | robust << 1 | damped.
// iters_since_relin (gbp.py:249) is not stored: the reference adds one to it for EVERY factor in every relinearise_factors() call
// (gbp.py:79-80) and zeroes it when the factor relinearises, so it is  (graph clock) - (clock value at the factor's last zero),  and the
// word holds the latter (20 bits, modulo).  A factor that does nothing but age -- every factor of a steady sweep -- keeps its word, and
// the sweep writes no state at all (rounds 1-3 stored the count itself: 4 bytes per factor and sweep written, on a 16-byte stride).
// The clock (Params::clk) advances once per call that runs the relinearisation test; ages saturate at ITERS_MAX like the stored
// counter did (only `>= min_linear` and `== num_undamped` are ever tested, both far below the cap, checked at create).
// "rank" (< 64) is constant per factor: its index among the same-camera factors of its tile (fused sweep).
// "pending": the factor has been told to linearise again at the belief means (relinearise_factors / compute_all_factors called on
// their own, gbp.py:60-80) but its messages have not been recomputed since.  A message is stored as coefficients in the rows of the
// Jacobian at the stored linearisation point, so the point moves when the messages are next computed -- at the belief means, which
// cannot change before that (beliefs are sums of messages) -- and the views show the belief means as the linearisation point meanwhile.
constexpr int STATE_SHIFT = 12;
constexpr unsigned CLK_MASK = (1u << (32 - STATE_SHIFT)) - 1u;
constexpr int ITERS_MAX = (1 << (31 - STATE_SHIFT)) - 1;     // iters_since_relin saturates here (524 287)
constexpr unsigned STATE_RANK_MASK = 0x1ffu;
constexpr int STATE_PENDING = 1 << 11;
GBP_HD int state_age(int st, int clk) { return (int)(((unsigned)clk - ((unsigned)st >> STATE_SHIFT)) & CLK_MASK); }     // iters_since_relin at clock `clk`
GBP_HD int state_rank(int st) { return (st >> 2) & (int)STATE_RANK_MASK; }
GBP_HD int state_pack(int iters, int clk, int rank, bool robust, bool damped, bool pending = false)      // a factor whose age is `iters` at clock `clk`
{
    return (int)(((((unsigned)clk - (unsigned)iters) & CLK_MASK) << STATE_SHIFT) | (pending ? (unsigned)STATE_PENDING : 0u) | ((unsigned)rank << 2) |
                 (robust ? 2u : 0u) | (damped ? 1u : 0u));
}
GBP_HD int state_set_age(int st, int iters, int clk)
{
    return (int)(((((unsigned)clk - (unsigned)iters) & CLK_MASK) << STATE_SHIFT) | ((unsigned)st & ((1u << STATE_SHIFT) - 1u)));
}

// Per-factor front of FactorGraph.synchronous_iteration (gbp.py:86-92): robustify (gbp.py:296-332), relinearisation
// test (gbp.py:64-80) and damping switch (gbp.py:50-51).  Returns true when the factor must relinearise at the belief
// means; `d` is the eta damping for this sweep.  x0 is NOT modified here (the old point is still needed to rebuild the
// factor's old messages).
template <int LOSS>
GBP_HD bool factor_decide(const Params &p, const double (&x0)[9], const double (&z)[2], int &st, double &avar,
                           const double (&muC)[6], const double (&muL)[3], double &d)
{
    int iters = state_age(st, p.clk - p.clk_inc);           // iters_since_relin before this call (state word header)
    bool robust = (st & 2) != 0, damped = (st & 1) != 0;
    const bool pending = (st & STATE_PENDING) != 0;          // told to relinearise earlier (stage-wise calls): the point moves now
    if (LOSS != 0 && p.robustify) {
        double h0[2];
        if (pending) {                                       // (its linearisation point is the belief means already: state word header)
            const double xm[9] = {muC[0], muC[1], muC[2], muC[3], muC[4], muC[5], muL[0], muL[1], muL[2]};
            project(xm, p.K, h0);
        } else {
            project(x0, p.K, h0);
        }
        avar = robust_variance(LOSS, p.sigma2, p.nstds, z[0] - h0[0], z[1] - h0[1], robust);
    } else if (p.robustify) {
        avar = p.sigma2;                       // loss None: adaptive = gauss_noise_var  gbp.py:302-303
    }
    bool relin = false;
    if (p.local_relin && !(p.stage & STAGE_NO_TEST)) {
        double d2 = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) d2 += (x0[i] - muC[i]) * (x0[i] - muC[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) d2 += (x0[6 + i] - muL[i]) * (x0[6 + i] - muL[i]);
        // (a pending factor's linearisation point IS the belief means already, as far as the reference can tell: distance 0)
        if (!pending && sqrt(d2) > p.beta && iters >= p.min_linear) {
            iters = 0;
            damped = false;
            relin = true;
        } else {
            // saturating (state word header): a converged factor that never relinearises behaves like the reference's unbounded
            // Python int for ever
            iters = min(iters + 1, ITERS_MAX);
        }
    }
    if (p.local_relin && iters == p.num_undamped) damped = true;      // gbp.py:50-51 (equality, not >=)
    relin = relin || pending;
    d = p.local_relin ? (damped ? p.eta_damping : 0.0) : p.eta_damping;   // gbp.py:52-54
    st = state_pack(iters, p.clk, state_rank(st), robust, damped);      // (the word of a factor that has only aged is unchanged)
    return relin;
}

// Factor.compute_factor in compact form (gbp.py:267-294): J = [Jc | Jl], rho = J x0 + z - h(x0), s = 1 / adaptive var
// (factor_geometry: the part that needs nothing but the linearisation point and the measurement)
GBP_HD void factor_geometry(const Params &p, const double (&x0)[9], const double (&z)[2], Lin &L)
{
    double h[2];
    linearise(x0, p.K, L.Jc, L.Jl, h);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) acc += L.Jc[r][i] * x0[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc += L.Jl[r][i] * x0[6 + i];
        L.rho[r] = acc + z[r] - h[r];
    }
}
GBP_HD void factor_linearise(const Params &p, const double (&x0)[9], const double (&z)[2], double avar, double d, Lin &L)
{
    L.d = d;
    L.s = rcp(avar);
    factor_geometry(p, x0, z, L);
}

// One factor's sweep in registers (gbp.py:82-84, 64-80, 46-54, 334-373), covariance form (gbp_math.hpp header).
//   in : x0, z, state, adaptive variance,
//        muC | PC = mean | covariance of the camera belief, muL = mean of the landmark belief, load_PL(PL) yields its covariance,
//        old message coefficients qC / qL and old cores WC / VL
//   out: new qC / qL / WC / VL (both messages from the OLD ones, gbp.py:371-373), dense new etas eCn / eLn and Lambdas
//        MCn / MLn for the belief sums, state / avar updated; returns true when the factor relinearised: store_x0(x0) has then been
//        called with the new linearisation point.  (x0, muC, PC, muL are scratch.)
//   XTRA: xt = this slot's dense remainder (read and updated), see Params::xtra.
template <int LOSS, bool XTRA, typename LmkCov, typename StoreX0>
GBP_HD bool factor_core(const Params &p, double (&x0)[9], const double (&z)[2], int &st, double &avar,
                        double (&muC)[6], double (&PC)[21], double (&muL)[3], LmkCov &&load_PL, StoreX0 &&store_x0,
                        double (&qC)[2], double (&qL)[2], double (&WC)[3], double (&VL)[3],
                        double (&eCn)[6], double (&eLn)[3], double (&MCn)[21], double (&MLn)[6], double *xt = nullptr, const Lin *geom = nullptr)
{
    // The factor's geometry at its stored linearisation point -- rotation, projection, the 2x9 Jacobian, rho: a quarter of the
    // arithmetic -- needs nothing of the beliefs: it comes FIRST, so that it runs while the camera record, the last thing the sweep
    // gathers, is still on its way (the relinearisation test below is the first instruction to read it; round 5 ran that test first
    // and every wave sat out the gather's whole round trip).  `geom`: the caller has made it already (fused sweep, PINNED variant).
    Lin L;
    if (geom) {
        L = *geom;
    } else {
        factor_geometry(p, x0, z, L);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(L.rho[0]), "+v"(L.rho[1]));      // (the order is the point: keep the scheduler from sinking it behind the wait)
#endif
    }
    double d;
    const bool relin = factor_decide<LOSS>(p, x0, z, st, avar, muC, muL, d);
    L.d = d;
    L.s = rcp(avar);
    double PL[6];
    load_PL(PL);
    double dqC[2] = {d * qC[0], d * qC[1]}, dqL[2] = {d * qL[0], d * qL[1]};      // the old etas' share of the new ones (gbp.py:368)
    double xn[XTRA ? XTRA_ROW : 1];
    // A factor that keeps its linearisation point -- every factor of most sweeps -- eliminates straight from the beliefs: its old
    // message has the Jacobian of its new block, so belief - old message + own block is ONE rank-2 update (Q = s I - W).  One that
    // relinearises (or carries a dense remainder) takes the old message out of the beliefs first, with the old Jacobian (in place:
    // mu | P become the cavity's), and then runs the same elimination with no old message at the new point.
    if (XTRA || relin) {
        if (relin) {                                       // gbp.py:75-78: the new linearisation point = the belief means
#pragma unroll
            for (int i = 0; i < 6; ++i) x0[i] = muC[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) x0[6 + i] = muL[i];
            store_x0(x0);                                  // at once: the stores' operands do not travel through the eliminations
        }
        downdate<3>(PL, muL, L.Jl[0], L.Jl[1], VL, qL);
        downdate<6>(PC, muC, L.Jc[0], L.Jc[1], WC, qC);
        if (XTRA) {
            // e_old = J_old^T q_old + x_old: the remainder leaves the cavity too (mu' -= P' x_old).  Damped in the very sweep it
            // relinearises: d e_old leaves the span of the new Jacobian and is carried densely; otherwise only the old remainder decays.
            double xo[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) xo[i] = xt[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < 6; ++j) a += PC[i <= j ? Sym<6>::at(i, j) : Sym<6>::at(j, i)] * xo[j];
                muC[i] -= a;
                xn[i] = relin ? d * (jcomb<6>(i, qC[0], qC[1], L.Jc[0], L.Jc[1]) + xo[i]) : d * xo[i];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) a += PL[i <= j ? Sym<3>::at(i, j) : Sym<3>::at(j, i)] * xo[6 + j];
                muL[i] -= a;
                xn[6 + i] = relin ? d * (L.Jl[0][i] * qL[0] + L.Jl[1][i] * qL[1] + xo[6 + i]) : d * xo[6 + i];
            }
        }
        if (relin) {
            factor_linearise(p, x0, z, avar, d, L);
            // the in-span part of the old eta lives in the rows of the OLD Jacobian: it cannot be mixed into the new coefficients
            // (d is 0 here, or -- XTRA -- the old eta travels in the remainder xn)
            dqC[0] = 0.0; dqC[1] = 0.0; dqL[0] = 0.0; dqL[1] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { WC[k] = 0.0; VL[k] = 0.0; }
        qC[0] = 0.0; qC[1] = 0.0; qL[0] = 0.0; qL[1] = 0.0;
    }
    double Vn[3], Wn[3], rL[2], rC[2];
    eliminate<3>(PL, muL, L.Jl[0], L.Jl[1], VL, qL, L.rho, L.s, Wn, rC);      // landmark out: the message to the camera
    eliminate<6>(PC, muC, L.Jc[0], L.Jc[1], WC, qC, L.rho, L.s, Vn, rL);      // camera out: the message to the landmark
    qL[0] = (1.0 - d) * rL[0] + dqL[0]; qL[1] = (1.0 - d) * rL[1] + dqL[1];
    qC[0] = (1.0 - d) * rC[0] + dqC[0]; qC[1] = (1.0 - d) * rC[1] + dqC[1];
#pragma unroll
    for (int k = 0; k < 3; ++k) { VL[k] = Vn[k]; WC[k] = Wn[k]; }
    dense_message<3>(L.Jl[0], L.Jl[1], qL, VL, eLn, MLn);
    dense_message<6>(L.Jc[0], L.Jc[1], qC, WC, eCn, MCn);
    if (XTRA) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { eCn[i] += xn[i]; xt[i] = xn[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { eLn[i] += xn[6 + i]; xt[6 + i] = xn[6 + i]; }
    }
    return relin;
}

// ba.py:96-99 counts the factors with iters_since_relin == 0 after every sweep; the sweep kernels add their own count into a
// per-sweep counter instead.  The counter is RELIN_LANES words wide and a wave adds to word (workgroup mod RELIN_LANES): one
// word for everybody serialised 16 k same-address atomics in L2 and DOUBLED the sweep time whenever a few per cent of the
// factors relinearised (measured: 0.116 -> 0.226 ms).
constexpr int RELIN_LANES = 64;
GBP_DEV int relin_in_wave(bool relin) { return __popcll(__ballot(relin)); }     // call with the whole tile's lanes active or not: exec-masked
GBP_DEV void relin_add(const Params &p, int n)                                   // one lane of the wave
{
    if (p.relin_slot && n) atomicAdd(p.relin_slot + (blockIdx.x & (RELIN_LANES - 1)), n);
}

// The dense messages of a slot (for the belief sums of the general path and the parity views): etas and precisions are
// rebuilt from their coefficients / cores at the stored linearisation point.
GBP_DEV void dense_messages(const Params &p, int slot, double (&eC)[6], double (&MC)[21], double (&eL)[3], double (&ML)[6]);

// the part of a camera record a factor reads: mean | covariance, 14 x 16 bytes
GBP_DEV void load_cam_record(const double *__restrict__ rec, double (&mu)[6], double (&cov)[21])
{
    const double2 *r2 = reinterpret_cast<const double2 *>(rec);
    double v[CAMHEAD];
#pragma unroll
    for (int i = 0; i < CAMHEAD / 2; ++i) { const double2 t = r2[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
#pragma unroll
    for (int i = 0; i < 6; ++i) mu[i] = v[CAM_MU + i];
#pragma unroll
    for (int i = 0; i < 21; ++i) cov[i] = v[CAM_COV + i];
}

// A camera belief from its eta | Lambda (v, the same 27 values in every participating lane), written in both forms.  Lanes 0..6 of
// a wave: lane c < 6 solves Lambda x = e_c (column c of the covariance), lane 6 solves Lambda mu = eta -- seven lanes instead of one
// lane doing seven solves in a row (VariableNode.update_belief gbp.py:189-193, plus the inverse the factors read, gbp_math.hpp).
GBP_DEV void cam_belief_store(const double (&v)[27], double *__restrict__ rec, int lane)
{
    if (lane >= 7) return;
    double lam[21], invd[6], e[6];
#pragma unroll
    for (int k = 0; k < 21; ++k) lam[k] = v[6 + k];
    ldl_factor<6>(lam, invd);
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] = lane == 6 ? v[i] : (i == lane ? 1.0 : 0.0);
    ldl_forward<6>(lam, e);
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] *= invd[i];
    ldl_backward<6>(lam, e);
    if (lane == 6) {
        double2 *r2 = reinterpret_cast<double2 *>(rec + CAM_MU);
        r2[0] = make_double2(e[0], e[1]); r2[1] = make_double2(e[2], e[3]); r2[2] = make_double2(e[4], e[5]);
        rec[CAM_COV + 21] = 0.0;
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i <= lane) rec[CAM_COV + i * 6 - (i * (i - 1)) / 2 + (lane - i)] = e[i];
    }
}

// slot -> (valid, camera, landmark) through the tile table and the meta word
GBP_DEV bool slot_info(const Params &p, int slot, int &cam, int &lmk)
{
    const int4 td = p.tiles[slot >> 6];
    if ((slot & 63) >= td.z) return false;
    const unsigned m = slot_meta(p, slot);
    cam = (int)(m >> META_LMK_BITS);
    lmk = td.x + (int)(m & ((1u << META_LMK_BITS) - 1u));
    return true;
}

GBP_DEV void dense_messages(const Params &p, int slot, double (&eC)[6], double (&MC)[21], double (&eL)[3], double (&ML)[6])
{
    double x0[9], Jc[2][6], Jl[2][3], h[2], qC[2], qL[2], WC[3], VL[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) x0[k] = p.lin[lin_at(slot, ROW_X0 + k)];
    linearise(x0, p.K, Jc, Jl, h);
#pragma unroll
    for (int k = 0; k < 2; ++k) { qC[k] = p.msg[msg_at(slot, ROW_QC + k)]; qL[k] = p.msg[msg_at(slot, ROW_QL + k)]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) WC[k] = p.msg[msg_at(slot, ROW_WC + k)];
#pragma unroll
    for (int k = 0; k < 3; ++k) VL[k] = p.msg[msg_at(slot, ROW_VL + k)];
    dense_message<6>(Jc[0], Jc[1], qC, WC, eC, MC);
    dense_message<3>(Jl[0], Jl[1], qL, VL, eL, ML);
    if (p.xtra) {
#pragma unroll
        for (int k = 0; k < 6; ++k) eC[k] += p.xtra[(size_t)slot * XTRA_ROW + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) eL[k] += p.xtra[(size_t)slot * XTRA_ROW + 6 + k];
    }
}

GBP_DEV void wave_lds_sync()
{
    // all LDS traffic of this wave issued so far has completed; nothing may be moved across
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// A landmark belief from its eta | Lambda (b), in the form the factors read it: mean | covariance (VariableNode.update_belief
// gbp.py:189-193 plus the inverse, gbp_math.hpp)
GBP_DEV void lmk_belief_store(const double (&b)[9], double *__restrict__ lr)
{
    double eta[3] = {b[0], b[1], b[2]}, lam[6] = {b[3], b[4], b[5], b[6], b[7], b[8]}, mu[3], sig[6];
    spd_solve_inverse<3>(lam, eta, mu, sig);
    double2 *d0 = reinterpret_cast<double2 *>(lr + LR_MU);
    d0[0] = make_double2(mu[0], mu[1]); d0[1] = make_double2(mu[2], sig[0]);
    d0[2] = make_double2(sig[1], sig[2]); d0[3] = make_double2(sig[3], sig[4]);
    lr[LR_COV + 5] = sig[5];                                 // (the double behind it holds the landmark's slot range)
}

// What the belief phase of a tile needs besides the messages, fetched ahead of it: nine lanes per landmark add one belief entry
// each, seven landmarks per pass -- lane (g, k) = (lane / 9, lane % 9) holds prior entry k of landmark 7 b + g for pass b -- and
// lane l < nl holds landmark l's slot range relative to the tile (two bytes).
constexpr int LMK_PASSES = (TILE_LMKS + 6) / 7;
constexpr int ROWS_CONT = 1 << 16, ROWS_MORE = 1 << 17;     // flags beside the two 8-bit slot bounds of LmkPre::rows
struct LmkPre {
    double pri[LMK_PASSES];
    int rx, ry;           // lane l < nl: landmark l's slot range as it is stored (absolute slots).  RAW: what the belief phase needs of it
                          // (lmk_rows) is formed when that phase runs -- formed at the prefetch it made the wave wait for ALL its loads
                          // (s_waitcnt vmcnt(0)) at the top of every iteration, before the phase that is there to run while they fly
};
// the landmark's slots inside tile t, and whether it began before it (ROWS_CONT) / goes on behind it (ROWS_MORE)
GBP_DEV int lmk_rows(const LmkPre &q, int lane, int t, int nl)
{
    if (lane >= nl) return 0;
    const int a = q.rx - t * WTILE, b = q.ry - t * WTILE;
    return max(a, 0) | (min(b, WTILE) << 8) | (a < 0 ? ROWS_CONT : 0) | (b > WTILE ? ROWS_MORE : 0);
}
GBP_DEV void lmk_prefetch(const Params &p, int lane, int t, int l0, int nl, LmkPre &q)
{
    const int g = (lane * 57) >> 9, k = lane - g * 9;       // lane / 9 for lane < 64
    const double *base = p.lrec + (size_t)l0 * LREC;        // wave-uniform: one address register per access
#pragma unroll
    for (int b = 0; b < LMK_PASSES; ++b) {
        const int li = b * 7 + g;
        q.pri[b] = (g < 7 && li < nl) ? base[(unsigned)(li * LREC + LR_PRIOR + k)] : 0.0;
    }
    q.rx = 0; q.ry = 0;
    if (lane < nl) {
        const int2 r = *reinterpret_cast<const int2 *>(base + (unsigned)(lane * LREC + LR_ROWS));
        q.rx = r.x; q.ry = r.y;
    }
    (void)t;
}

// The landmark beliefs of a tile from the wave's LDS scratch wl = [64][9] new messages (eta 3 | Lambda 6 per factor lane): prior +
// messages in adj_factors order (gbp.py:182-193), then mean and covariance.  The sums go back into rows of wl that no later pass
// reads -- the first factor of a landmark of pass b sits in row 7 b or behind: every landmark in front of it has a factor, or the
// packer has made sure of it (landmarks without factors take no slot: gbp_build.hpp k_pack_next) -- and one lane per landmark
// solves and writes the record.  (One lane per landmark
// reading 9 doubles per message was the longest phase of the loop: 27 % of the wave-time with 6 of 64 lanes busy.)
GBP_DEV void tile_landmark_beliefs(const Params &p, double *wl, int lane, int t, int l0, int nl, const LmkPre &q)
{
    const int g = (lane * 57) >> 9, k = lane - g * 9;
    const int my_rows = lmk_rows(q, lane, t, nl);
#pragma unroll
    for (int b = 0; b < LMK_PASSES; ++b) {
        if (b * 7 >= nl) break;                             // wave-uniform
        const int li = b * 7 + g;
        const int rows = __shfl(my_rows, li, 64);
        if (g < 7 && li < nl) {
            // a landmark that spans tiles: the sum of THIS tile's part alone (the prior joins in k_lmk_finish_parts)
            double acc = (rows & (ROWS_CONT | ROWS_MORE)) ? 0.0 : q.pri[b];
            int r = rows & 0xff;
            const int r1 = (rows >> 8) & 0xff;
            for (; r + 4 <= r1; r += 4) {                   // four reads in flight, the additions in adj_factors order as before
                const double v0 = wl[r * 9 + k], v1 = wl[(r + 1) * 9 + k], v2 = wl[(r + 2) * 9 + k], v3 = wl[(r + 3) * 9 + k];
                acc += v0; acc += v1; acc += v2; acc += v3;
            }
            for (; r < r1; ++r) acc += wl[r * 9 + k];
            wl[li * 9 + k] = acc;
        }
    }
    wave_lds_sync();
    if (lane < nl) {
        double b[9];
#pragma unroll
        for (int k2 = 0; k2 < 9; ++k2) b[k2] = wl[lane * 9 + k2];
        if (my_rows & (ROWS_CONT | ROWS_MORE)) {            // (at most the first and the last landmark of a tile)
            double *dst = p.parts + ((size_t)2 * t + ((my_rows & ROWS_CONT) ? 0 : 1)) * PART_ROW;
#pragma unroll
            for (int k2 = 0; k2 < 9; ++k2) dst[k2] = b[k2];
        } else {
            lmk_belief_store(b, p.lrec + (size_t)(l0 + lane) * LREC);
        }
    }
}

// Beliefs of the landmarks that span tiles, after a sweep: thread t looks at the FIRST landmark of tile t; if it began in an earlier
// tile and ends in this one, the thread adds up prior + the parts of all its tiles in tile (= adj_factors) order -- row 2 t0 + 1 of
// the tile it begins in, rows 2 t' of the tiles it continues in (tile_landmark_beliefs) -- and writes mean | covariance
// (VariableNode.update_belief gbp.py:176-198).  At most one landmark ends that way per tile; no list of them is needed.
GBP_DEV void finish_landmark_parts(const Params &p, int t)
{
    const int4 td = p.tiles[t];
    if (td.y < 1 || td.z < 1) return;
    double *lr = p.lrec + (size_t)td.x * LREC;
    const int2 rows = *reinterpret_cast<const int2 *>(lr + LR_ROWS);
    if (rows.x >= t * WTILE || rows.y > (t + 1) * WTILE) return;      // did not begin earlier, or goes on: somebody else's
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = lr[LR_PRIOR + k];
    const int t0 = rows.x / WTILE;
    for (int tt = t0; tt <= t; ++tt) {
        const double *src = p.parts + ((size_t)2 * tt + (tt == t0 ? 1 : 0)) * PART_ROW;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += src[k];
    }
    lmk_belief_store(acc, lr);
}

// ------------------------------------------------------------------ general sweep, stage 2 --
// VariableNode.update_belief for one landmark (gbp.py:176-198): prior + messages in adj_factors order
// (= ascending reference factor id = slot order inside the landmark), then mu = Lambda^-1 eta.
GBP_DEV void landmark_sum_from_hbm(const Params &p, int l, double (&acc)[9])
{
    const double *lr = p.lrec + (size_t)l * LREC;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = lr[LR_PRIOR + k];
    const int2 rows = *reinterpret_cast<const int2 *>(lr + LR_ROWS);
    for (int s = rows.x; s < rows.y; ++s) {
        double eC[6], MC[21], eL[3], ML[6];
        dense_messages(p, s, eC, MC, eL, ML);
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += eL[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[3 + k] += ML[k];
    }
}

GBP_DEV void landmark_belief_from_hbm(const Params &p, int l)
{
    double acc[9];
    landmark_sum_from_hbm(p, l, acc);
    lmk_belief_store(acc, p.lrec + (size_t)l * LREC);
}

// ------------------------------------------------------------------ peer-store camera exchange --
// Landmark-sharded sweep without a collective call: every rank owns a MAILBOX in its own
// device memory -- two sweep-parity halves of [n_ranks][C] rows of 28 doubles: a camera's 27 partial sums + pad -- and the
// kernel that produces a rank's camera partial sums stores each row straight into the mailbox of EVERY rank (xGMI peer stores
// on a multi-GPU node); whoever finishes camera c polls the n_ranks rows c in its own mailbox half and adds the parts in rank order.
// Two halves are enough: a rank can write sweep k+2's sums only after its own finish of sweep k+1, which needs every peer's sums of
// sweep k+1, which a peer produces after ITS finish of sweep k.
//
// ONE trip per exchange (round 6).  THE DATA IS ITS OWN ARRIVAL FLAG: an empty mailbox slot holds PEER_EMPTY -- a quiet NaN with a
// payload no arithmetic produces -- the sender just stores its 27 doubles (each an 8-byte single-copy-atomic store, fire and forget),
// the finisher's poll IS its data load (repeat until no slot of the row is empty), and the slots it has read are emptied again ONE
// EXCHANGE LATER, at the start of the next push (peer_push_row) -- by the two-halves argument above long before any peer stores into them
// again (that peer must first finish the exchange this push belongs to, and run a sweep).  Rounds 3-5 stored the row, waited for the stores' acknowledgements (s_waitcnt vmcnt(0): data
// before tag), raised a tag, and the finisher polled the tag and THEN loaded the row: four dependent memory round trips per exchange --
// across xGMI each is microseconds -- where this has two (the stores' flight, the poll that finds them), and no ordering between
// different addresses is asked of the link at all.
//
// No fences: a system-scope release / acquire fence on gfx950 writes back / invalidates a whole L2 (8000 waves doing that after the
// sweep cost 120 us per reduce launch).  Every mailbox access is itself a system-scope relaxed atomic -- write-through stores, loads
// that bypass the caches (/opt/skills/guides/MI355X_MICROARCH.md).
constexpr int MAX_PEERS = 16;
constexpr int PEER_ROW = 28;                          // doubles per mailbox row: 27 sums | pad (rows start on 16 bytes)
constexpr unsigned PEER_EMPTY32 = 0x7ff87ff8u;        // both halves of PEER_EMPTY (the mailbox is filled with hipMemsetD32Async)
constexpr unsigned long long PEER_EMPTY = ((unsigned long long)PEER_EMPTY32 << 32) | PEER_EMPTY32;      // a quiet NaN with a payload: arithmetic makes 0x7ff8000000000000 (or its negative), and it propagates payloads only from operands that carry one
constexpr int PEER_CHUNK = 8;                         // rows of a camera polled together (one trip for up to eight ranks)
struct PeerOut {
    int n;                                            // ranks (0: no peer stores)
    int C;                                            // cameras (rows per rank block)
    double *dst[MAX_PEERS];                           // rank r's mailbox half of this sweep, the block of THIS rank: [C][PEER_ROW]
    unsigned long long seq;                           // exchange number (1, 2, ...): the self-test's probe values depend on it
    double *stale;                                    // THIS rank's mailbox half of the PREVIOUS exchange: [n][C][PEER_ROW], read then, emptied now
};
struct PeerWait {
    double *src;                                      // this rank's mailbox half: [n_parts][C][PEER_ROW], or NULL (parts are plain arrays)
    unsigned long long seq;
    long long timeout_ticks;                          // wall_clock64 ticks (100 MHz): a peer that never arrives must not hang the GPU
    int *err;                                         // set to 1 on time-out (reported by gbp_ba_sync)
    unsigned long long *clk;                          // instrumented runs: where workgroup 0 stores its start time, or NULL
};

GBP_DEV void peer_store(double *dst, double v) { __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
GBP_DEV double peer_load(const double *src) { return __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
GBP_DEV bool peer_is_empty(double v) { return (unsigned long long)__double_as_longlong(v) == PEER_EMPTY; }
GBP_DEV double peer_empty_value() { return __longlong_as_double((long long)PEER_EMPTY); }

// ONE wave: lanes 0..26 hold camera c's partial sums; row c of this rank's block in every mailbox gets them.  Nothing to wait for.
// The same wave first empties the rows c of the OTHER half of its own mailbox -- what the previous exchange delivered and this rank's
// finish has read, one kernel ago -- for the exchange after this one.  (Emptied by the finisher right after its read, the write-through
// resets were the last stores of the launch and the kernel boundary behind it waited for their round trip: 5.7 us from the last
// workgroup to the next sweep's start where the plain reduce has 3.2, tools/boundary_probe.py.  Here they have the whole launch to land.)
// A peer stores into those slots again only after its finish of THIS exchange, which needs the push below, and a sweep of its own.
GBP_DEV void peer_push_row(const PeerOut &peer, int c, double v, int lane)
{
    if (peer.stale)
        for (int r = 0; r < peer.n; ++r)
            if (lane < 27) peer_store(peer.stale + ((size_t)r * peer.C + c) * PEER_ROW + lane, peer_empty_value());
    for (int r = 0; r < peer.n; ++r)
        if (lane < 27) peer_store(peer.dst[r] + (size_t)c * PEER_ROW + lane, v);
}

// ONE wave: entry `lane` (< 27) of rows c of parts r0 .. r0 + PEER_CHUNK - 1 (those below n_parts) of the mailbox half `src`
// ([n_parts][C][PEER_ROW]), polled until none of them is empty (they are emptied again one exchange later: peer_push_row).  false on time-out.
GBP_DEV bool peer_take_rows(const PeerWait &wait, int n_parts, int C, int c, int r0, int lane, double (&v)[PEER_CHUNK], long long t0)
{
    bool ok = true;
    for (;;) {
        bool missing = false;
#pragma unroll
        for (int j = 0; j < PEER_CHUNK; ++j) {
            v[j] = 0.0;
            if (lane < 27 && r0 + j < n_parts) {
                v[j] = peer_load(wait.src + ((size_t)(r0 + j) * C + c) * PEER_ROW + lane);
                missing = missing || peer_is_empty(v[j]);
            }
        }
        if (!__any(missing)) break;
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > wait.timeout_ticks) { ok = false; break; }
    }
    ok = __all(ok);
    if (!ok) {
        if (lane == 0) atomicExch(wait.err, 1);
        return false;
    }
    return true;
}

// belief_c = prior_c + sum over parts (fixed order) of part_r[c]; mu_c = Lambda^-1 eta.  One wavefront per camera: lane k < 27
// adds entry k of the parts (coalesced 216-byte rows) in rank order, seven lanes solve the 6x6.
// With wait.src the parts are rows of a mailbox half: the camera's wave takes them as they arrive (peer_take_rows).
// one WAVE finishes camera c (shared by k_cam_finish and the merged reduce-exchange-finish kernel of gbp_fused.hpp)
GBP_DEV void cam_finish_wave(const Params &p, const double *gathered, int n_parts, size_t part_stride, const PeerWait &wait, int c, int lane)
{
    double acc = 0.0;
    if (wait.src) {                                         // the parts are rows of this rank's mailbox half
        if (lane < 27) acc = p.cprior[(size_t)c * 27 + lane];
        const long long t0 = wall_clock64();
        for (int r0 = 0; r0 < n_parts; r0 += PEER_CHUNK) {
            double v[PEER_CHUNK];
            if (!peer_take_rows(wait, n_parts, p.C, c, r0, lane, v, t0)) break;      // (time-out: reported by gbp_ba_sync, the belief is garbage)
#pragma unroll
            for (int j = 0; j < PEER_CHUNK; ++j)
                if (r0 + j < n_parts) acc += v[j];         // rank order
        }
    } else if (lane < 27) {
        acc = p.cprior[(size_t)c * 27 + lane];
        for (int r = 0; r < n_parts; ++r) acc += gathered[(size_t)r * part_stride + (size_t)c * 27 + lane];
    }
    if (lane < 27) p.cbelief[(size_t)c * CBEL + lane] = acc;
    double v[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = __shfl(acc, k, 64);
    cam_belief_store(v, p.cbel + (size_t)c * CAMREC, lane);
}

// Factor.linpoint as the reference would show it: the stored point, or the belief means for a factor whose relinearisation is pending
GBP_DEV void effective_linpoint(const Params &p, int slot, double (&x0)[9])
{
    int cam, lmk;
    if ((slot_state(p, slot) & STATE_PENDING) && slot_info(p, slot, cam, lmk)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) x0[k] = p.cbel[(size_t)cam * CAMREC + CAM_MU + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) x0[6 + k] = p.lrec[(size_t)lmk * LREC + LR_MU + k];
        return;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) x0[k] = p.lin[lin_at(slot, ROW_X0 + k)];
}

}  // namespace gbp
