// gbp_kernels.hpp -- HIP kernels of the GBP bundle-adjustment sweep for gfx950 (MI355X).
//
// Data layout in HBM (all fp64 unless noted; "packed" = upper triangle row-major):
//   factors, INTERNAL order = landmark-major (stable by reference factor id inside a landmark),
//   structure-of-arrays with stride Fp (F rounded up to 256) so lane i of a wave touches
//   consecutive 8-byte words:
//       x0[9][Fp]   linearisation point (t, w, y)            Factor.linpoint        gbp.py:231
//       z[2][Fp]    measurement                              Factor.measurement     gbp.py:233
//       mc[27][Fp]  message to the camera  (eta 6 | Lambda 21 packed)  Factor.messages[0]
//       ml[9][Fp]   message to the landmark (eta 3 | Lambda 6 packed)  Factor.messages[1]
//       avar[Fp]    adaptive noise variance (only when loss != none)   gbp.py:242
//       fcam[Fp]    int32 camera index;  state[Fp] int32 = iters_since_relin<<12 | rank<<2 | robust<<1 | damped
//   landmarks, SoA stride Lp:  lbel[9][Lp] (eta 3 | Lambda 6), lmu[3][Lp], lprior[9][Lp];
//       lptr[L+1] = first internal factor of each landmark (its factors are contiguous)
//   cameras, array-of-records (gathered per factor, L2 resident: 500 cams = 136 KB):
//       cbel[C][34] = mu 6 | eta 6 | Lambda 21 | pad;  cprior[C][27] = eta 6 | Lambda 21
//       cptr[C+1], cadj[F] = internal factor ids of each camera in reference order
//
// Lambda_f / eta_f (90 doubles per factor in the reference) are never stored: they are rebuilt
// from x0 and z every sweep (2x9 Jacobian = ~150 flops vs 720 bytes of traffic).
//
// General sweep = k_factor (one lane per factor) -> k_lmk_belief (one lane per landmark over its
// contiguous messages) -> k_cam_partial (one block per camera, gather) -> k_cam_finish.
#pragma once
#include "gbp_math.hpp"

namespace gbp {

constexpr int CAMREC = 34;       // doubles per camera belief record: mu 6 | eta 6 | Lambda 21 | pad
constexpr int CAM_MU = 0, CAM_ETA = 6, CAM_LAM = 12;
constexpr int BLOCK = 256;

struct Params {
    int F, Fp, L, Lp, C;
    Intrinsics K;
    double sigma2, nstds, beta, eta_damping;
    int num_undamped, min_linear, loss;
    int robustify, local_relin;
    // factors
    double *x0, *z, *mc, *ml, *avar;
    int *fcam, *flmk, *state;
    // landmarks
    double *lbel, *lmu, *lprior;
    const int *lptr;
    // cameras
    double *cbel, *cprior;
    const int *cptr, *cadj;
};

// state word: iters_since_relin << 12 | rank << 2 | robust << 1 | damped.  "rank" (10 bits) is constant per
// factor: its index among the same-camera factors of its tile (fused sweep); the general kernels carry it along.
constexpr int STATE_SHIFT = 12;
constexpr unsigned STATE_RANK_MASK = 0x3ffu;
GBP_DEV int state_iters(int st) { return st >> STATE_SHIFT; }
GBP_DEV int state_rank(int st) { return (st >> 2) & (int)STATE_RANK_MASK; }
GBP_DEV int state_pack(int iters, int rank, bool robust, bool damped)
{
    return (int)(((unsigned)iters << STATE_SHIFT) | ((unsigned)rank << 2) | (robust ? 2u : 0u) | (damped ? 1u : 0u));
}

// Per-factor front half of FactorGraph.synchronous_iteration (gbp.py:86-92): robustify (gbp.py:296-332)
// -> relinearise test (gbp.py:64-80) -> damping switch (gbp.py:50-51) -> linearisation at the chosen point.
// Returns true when the factor relinearised (x0 was replaced by the belief means).
template <int LOSS>
GBP_DEV bool factor_prepare(const Params &p, double (&x0)[9], const double (&z)[2], int &st, double &avar,
                            const double (&muC)[6], const double (&muL)[3], Lin &L)
{
    int iters = state_iters(st);
    bool robust = (st & 2) != 0, damped = (st & 1) != 0;

    if (LOSS != 0 && p.robustify) {
        double h0[2];
        project(x0, p.K, h0);
        avar = robust_variance(LOSS, p.sigma2, p.nstds, z[0] - h0[0], z[1] - h0[1], robust);
    } else if (p.robustify) {
        avar = p.sigma2;                       // loss None: adaptive = gauss_noise_var  gbp.py:302-303
    }

    bool relinearised = false;
    if (p.local_relin) {
        double d2 = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) d2 += (x0[i] - muC[i]) * (x0[i] - muC[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) d2 += (x0[6 + i] - muL[i]) * (x0[6 + i] - muL[i]);
        if (sqrt(d2) > p.beta && iters >= p.min_linear) {
#pragma unroll
            for (int i = 0; i < 6; ++i) x0[i] = muC[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) x0[6 + i] = muL[i];
            iters = 0;
            damped = false;
            relinearised = true;
        } else {
            iters += 1;
        }
        if (iters == p.num_undamped) damped = true;      // gbp.py:50-51 (equality, not >=)
    }
    L.d = p.local_relin ? (damped ? p.eta_damping : 0.0) : p.eta_damping;   // gbp.py:52-54
    L.s = 1.0 / avar;

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
    st = state_pack(iters, state_rank(st), robust, damped);
    return relinearised;
}

// prepare + both messages (Factor.compute_messages gbp.py:334-373: both from the OLD messages, committed together)
template <int LOSS>
GBP_DEV void factor_step(const Params &p, double (&x0)[9], const double (&z)[2], int &st, double &avar,
                         const double (&etaC)[6], const double (&lamC)[21], const double (&muC)[6],
                         const double (&etaL)[3], const double (&lamL)[6], const double (&muL)[3],
                         double (&eC)[6], double (&MC)[21], double (&eL)[3], double (&ML)[6], bool &relinearised)
{
    Lin L;
    relinearised = factor_prepare<LOSS>(p, x0, z, st, avar, muC, muL, L);
    double eLn[3], MLn[6], MCn[21];
    message_to_landmark(L, etaC, lamC, eC, MC, eL, eLn, MLn);
    message_to_camera(L, etaL, lamL, eL, ML, eC, MCn);
#pragma unroll
    for (int i = 0; i < 3; ++i) eL[i] = eLn[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) ML[i] = MLn[i];
#pragma unroll
    for (int i = 0; i < 21; ++i) MC[i] = MCn[i];
}

GBP_DEV void load_cam_record(const double *__restrict__ rec, double (&eta)[6], double (&lam)[21], double (&mu)[6])
{
    const double2 *r2 = reinterpret_cast<const double2 *>(rec);
    double v[34];
#pragma unroll
    for (int i = 0; i < 17; ++i) { const double2 t = r2[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
#pragma unroll
    for (int i = 0; i < 6; ++i) mu[i] = v[CAM_MU + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) eta[i] = v[CAM_ETA + i];
#pragma unroll
    for (int i = 0; i < 21; ++i) lam[i] = v[CAM_LAM + i];
}

// ------------------------------------------------------------------ general sweep, stage 1 --
template <int LOSS>
__global__ __launch_bounds__(BLOCK) void k_factor(Params p)
{
    const int f = blockIdx.x * BLOCK + threadIdx.x;
    if (f >= p.F) return;
    const size_t Fp = (size_t)p.Fp, Lp = (size_t)p.Lp;
    double x0[9], z[2], eC[6], MC[21], eL[3], ML[6];
#pragma unroll
    for (int k = 0; k < 9; ++k) x0[k] = p.x0[k * Fp + f];
    z[0] = p.z[f]; z[1] = p.z[Fp + f];
#pragma unroll
    for (int k = 0; k < 6; ++k) eC[k] = p.mc[k * Fp + f];
#pragma unroll
    for (int k = 0; k < 21; ++k) MC[k] = p.mc[(6 + k) * Fp + f];
#pragma unroll
    for (int k = 0; k < 3; ++k) eL[k] = p.ml[k * Fp + f];
#pragma unroll
    for (int k = 0; k < 6; ++k) ML[k] = p.ml[(3 + k) * Fp + f];
    int st = p.state[f];
    double avar = (LOSS != 0) ? p.avar[f] : p.sigma2;
    const int c = p.fcam[f], l = p.flmk[f];
    double etaC[6], lamC[21], muC[6], etaL[3], lamL[6], muL[3];
    load_cam_record(p.cbel + (size_t)c * CAMREC, etaC, lamC, muC);
#pragma unroll
    for (int k = 0; k < 3; ++k) etaL[k] = p.lbel[k * Lp + l];
#pragma unroll
    for (int k = 0; k < 6; ++k) lamL[k] = p.lbel[(3 + k) * Lp + l];
#pragma unroll
    for (int k = 0; k < 3; ++k) muL[k] = p.lmu[k * Lp + l];

    bool relin;
    factor_step<LOSS>(p, x0, z, st, avar, etaC, lamC, muC, etaL, lamL, muL, eC, MC, eL, ML, relin);

    if (relin) {
#pragma unroll
        for (int k = 0; k < 9; ++k) p.x0[k * Fp + f] = x0[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) p.mc[k * Fp + f] = eC[k];
#pragma unroll
    for (int k = 0; k < 21; ++k) p.mc[(6 + k) * Fp + f] = MC[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p.ml[k * Fp + f] = eL[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) p.ml[(3 + k) * Fp + f] = ML[k];
    p.state[f] = st;
    if (LOSS != 0) p.avar[f] = avar;
}

// ------------------------------------------------------------------ general sweep, stage 2 --
// VariableNode.update_belief for landmarks (gbp.py:176-198): prior + messages in adj_factors
// order (= ascending reference factor id = internal order inside the landmark), then mu.
__global__ __launch_bounds__(BLOCK) void k_lmk_belief(Params p)
{
    const int l = blockIdx.x * BLOCK + threadIdx.x;
    if (l >= p.L) return;
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

// One block per camera: sum of the messages of its factors (gathered through cadj), WITHOUT the
// prior, into partial[c][27].  Fixed shape reduction -> bitwise reproducible.
__global__ __launch_bounds__(BLOCK) void k_cam_partial(Params p, double *__restrict__ partial)
{
    __shared__ double red[BLOCK / 64][27];
    const int c = blockIdx.x;
    const size_t Fp = (size_t)p.Fp;
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    const int e1 = p.cptr[c + 1];
    for (int e = p.cptr[c] + threadIdx.x; e < e1; e += BLOCK) {
        const int f = p.cadj[e];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] += p.mc[k * Fp + f];
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

// belief_c = prior_c + sum over ranks (fixed order) of partial_r[c]; mu_c = Lambda^-1 eta.
__global__ __launch_bounds__(64) void k_cam_finish(Params p, const double *__restrict__ gathered, int n_parts,
                                                   size_t part_stride)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= p.C) return;
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = p.cprior[(size_t)c * 27 + k];
    for (int r = 0; r < n_parts; ++r) {
        const double *src = gathered + (size_t)r * part_stride + (size_t)c * 27;
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] += src[k];
    }
    double *rec = p.cbel + (size_t)c * CAMREC;
#pragma unroll
    for (int k = 0; k < 27; ++k) rec[CAM_ETA + k] = acc[k];
    double eta[6], lam[21], mu[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) eta[k] = acc[k];
#pragma unroll
    for (int k = 0; k < 21; ++k) lam[k] = acc[6 + k];
    spd_solve<6>(lam, eta, mu);
#pragma unroll
    for (int k = 0; k < 6; ++k) rec[CAM_MU + k] = mu[k];
    rec[33] = 0.0;
}

// ----------------------------------------------------------------------------- diagnostics --
// Factor.compute_residual at the current belief means (gbp.py:251-259); per-block partial sums of
// ||r|| (BAFactorGraph.are gbp_ba.py:61-69) and 0.5||r||^2/adaptive_var (FactorGraph.energy gbp.py:36-44).
__global__ __launch_bounds__(BLOCK) void k_residual(Params p, double *__restrict__ partials)
{
    __shared__ double red[BLOCK / 64][2];
    const int f = blockIdx.x * BLOCK + threadIdx.x;
    const size_t Fp = (size_t)p.Fp, Lp = (size_t)p.Lp;
    double nr = 0.0, en = 0.0;
    if (f < p.F) {
        const int c = p.fcam[f], l = p.flmk[f];
        double x[9], h[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] = p.cbel[(size_t)c * CAMREC + CAM_MU + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[6 + k] = p.lmu[k * Lp + l];
        project(x, p.K, h);
        const double r0 = h[0] - p.z[f], r1 = h[1] - p.z[Fp + f];
        nr = sqrt(r0 * r0 + r1 * r1);
        const double av = p.loss != 0 ? p.avar[f] : p.sigma2;
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

// ---------------------------------------------------------------------------------- set-up --
// np.max(factor.factor.lam) per factor at its current linearisation point (gbp_ba.py:31)
__global__ __launch_bounds__(BLOCK) void k_factor_lambda_max(Params p, double *__restrict__ fmax_out)
{
    const int f = blockIdx.x * BLOCK + threadIdx.x;
    if (f >= p.F) return;
    const size_t Fp = (size_t)p.Fp;
    double x0[9], Jc[2][6], Jl[2][3], h[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) x0[k] = p.x0[k * Fp + f];
    linearise(x0, p.K, Jc, Jl, h);
    const double av = p.loss != 0 ? p.avar[f] : p.sigma2;
    fmax_out[f] = factor_lambda_max(Jc, Jl, 1.0 / av);
}

// dense (eta_f 9, Lambda_f 81) of a range of factors for the parity views (Factor.factor gbp.py:230,292)
__global__ __launch_bounds__(BLOCK) void k_export_factors(Params p, const int *__restrict__ ids, int n,
                                                          double *__restrict__ eta_out, double *__restrict__ lam_out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int f = ids[i];
    const size_t Fp = (size_t)p.Fp;
    double x0[9], Jc[2][6], Jl[2][3], h[2], J[2][9], rho[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) x0[k] = p.x0[k * Fp + f];
    linearise(x0, p.K, Jc, Jl, h);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int k = 0; k < 6; ++k) J[r][k] = Jc[r][k];
#pragma unroll
        for (int k = 0; k < 3; ++k) J[r][6 + k] = Jl[r][k];
    }
    const double zz[2] = {p.z[f], p.z[Fp + f]};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += J[r][k] * x0[k];
        rho[r] = acc + zz[r] - h[r];
    }
    const double s = 1.0 / (p.loss != 0 ? p.avar[f] : p.sigma2);
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        eta_out[(size_t)i * 9 + a] = s * (J[0][a] * rho[0] + J[1][a] * rho[1]);
#pragma unroll
        for (int b = 0; b < 9; ++b) lam_out[(size_t)i * 81 + a * 9 + b] = s * (J[0][a] * J[0][b] + J[1][a] * J[1][b]);
    }
}

// Sigma = Lambda^-1 for the covariance view (VariableNode.Sigma gbp.py:192)
__global__ __launch_bounds__(BLOCK) void k_covariances(Params p, double *__restrict__ cam_sig, double *__restrict__ lmk_sig)
{
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v < p.C) {
        double lam[21], sig[21];
#pragma unroll
        for (int k = 0; k < 21; ++k) lam[k] = p.cbel[(size_t)v * CAMREC + CAM_LAM + k];
        spd_inverse<6>(lam, sig);
#pragma unroll
        for (int k = 0; k < 21; ++k) cam_sig[(size_t)v * 21 + k] = sig[k];
    } else if (v < p.C + p.L) {
        const int l = v - p.C;
        double lam[6], sig[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) lam[k] = p.lbel[(size_t)(3 + k) * p.Lp + l];
        spd_inverse<3>(lam, sig);
#pragma unroll
        for (int k = 0; k < 6; ++k) lmk_sig[(size_t)l * 6 + k] = sig[k];
    }
}

__global__ __launch_bounds__(BLOCK) void k_scale(double *__restrict__ a, size_t n, double factor)
{
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) a[i] *= factor;
}

__global__ __launch_bounds__(BLOCK) void k_fill_iters(int *__restrict__ state, int n, int iters)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) state[i] = (int)(((unsigned)iters << STATE_SHIFT) | ((unsigned)state[i] & ((1u << STATE_SHIFT) - 1u)));
}

}  // namespace gbp
