// gbp_view_kernels.hpp -- kernels behind the state views of include/gbp_ba.h (beliefs, messages, factors, relinearisation state, means export,
// eval_fn): launched by gbp_capi_views.hip only; each moves only the requested range
#pragma once
#include "gbp_kernels.hpp"

namespace gbp {

// VariableNode.belief (eta | Lambda) of every landmark as a view.  The sweep keeps a landmark belief in the form the factors read,
// mean | covariance; its information form is Lambda = Sigma^-1, eta = Lambda mu -- the belief as of the last update_belief, whatever has
// happened to messages or priors since (the stage-wise calls of gbp.py:46-84 change those without touching the beliefs).  The 3x3
// round trip costs ~cond(Lambda) * 1e-16 relative (1e-10 on the shipped data) against 72 bytes per landmark and sweep that no kernel reads.
__global__ __launch_bounds__(BLOCK) void k_lmk_belief_view(Params p, double *__restrict__ out)
{
    const int l = blockIdx.x * BLOCK + threadIdx.x;
    if (l >= p.L) return;
    const double *lr = p.lrec + (size_t)l * LREC;
    double sig[6], lam[6], mu[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) sig[k] = lr[LR_COV + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) mu[k] = lr[LR_MU + k];
    spd_inverse<3>(sig, lam);
    out[(size_t)l * 9 + 0] = lam[0] * mu[0] + lam[1] * mu[1] + lam[2] * mu[2];
    out[(size_t)l * 9 + 1] = lam[1] * mu[0] + lam[3] * mu[1] + lam[4] * mu[2];
    out[(size_t)l * 9 + 2] = lam[2] * mu[0] + lam[4] * mu[1] + lam[5] * mu[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) out[(size_t)l * 9 + 3 + k] = lam[k];
}


// dense (eta_f 9, Lambda_f 81) of a list of slots for the parity views (Factor.factor gbp.py:230,292)
__global__ __launch_bounds__(BLOCK) void k_export_factors(Params p, const int *__restrict__ slots, int n,
                                                          double *__restrict__ eta_out, double *__restrict__ lam_out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int slot = slots[i];
    double x0[9], Jc[2][6], Jl[2][3], h[2], J[2][9], rho[2];
    effective_linpoint(p, slot, x0);
    linearise(x0, p.K, Jc, Jl, h);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int k = 0; k < 6; ++k) J[r][k] = Jc[r][k];
#pragma unroll
        for (int k = 0; k < 3; ++k) J[r][6 + k] = Jl[r][k];
    }
    const double zz[2] = {p.lin[lin_at(slot, ROW_Z)], p.lin[lin_at(slot, ROW_Z + 1)]};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += J[r][k] * x0[k];
        rho[r] = acc + zz[r] - h[r];
    }
    const double s = 1.0 / slot_avar(p, slot);
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        eta_out[(size_t)i * 9 + a] = s * (J[0][a] * rho[0] + J[1][a] * rho[1]);
#pragma unroll
        for (int b = 0; b < 9; ++b) lam_out[(size_t)i * 81 + a * 9 + b] = s * (J[0][a] * J[0][b] + J[1][a] * J[1][b]);
    }
}

// linearisation point / measurement of a list of slots (Factor.linpoint gbp.py:231, Factor.measurement gbp.py:233)
__global__ __launch_bounds__(BLOCK) void k_export_lin(Params p, const int *__restrict__ slots, int n, double *__restrict__ x0_out,
                                                      double *__restrict__ z_out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int slot = slots[i];
    if (x0_out) {
        double x0[9];
        effective_linpoint(p, slot, x0);
#pragma unroll
        for (int k = 0; k < 9; ++k) x0_out[(size_t)i * 9 + k] = x0[k];
    }
    if (z_out) { z_out[(size_t)i * 2] = p.lin[lin_at(slot, ROW_Z)]; z_out[(size_t)i * 2 + 1] = p.lin[lin_at(slot, ROW_Z + 1)]; }
}

// relinearisation / robust state of a list of slots (gbp.py:242-249): iters_since_relin, flags (bit 0 damped, bit 1 robust),
// adaptive variance
__global__ __launch_bounds__(BLOCK) void k_export_relin(Params p, const int *__restrict__ slots, int n, int *__restrict__ iters,
                                                        unsigned char *__restrict__ flags, double *__restrict__ avar)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int slot = slots[i], st = slot_state(p, slot);
    if (iters) iters[i] = state_age(st, p.clk);
    if (flags) flags[i] = (unsigned char)(st & 3);
    if (avar) avar[i] = slot_avar(p, slot);
}

// iters_since_relin of a list of slots (ba.py:91-93 per factor), clamped to the counter's range
__global__ __launch_bounds__(BLOCK) void k_import_iters(Params p, const int *__restrict__ slots, int n, const int *__restrict__ iters)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int slot = slots[i];
    const int v = min(max(iters[i], 0), ITERS_MAX);
    set_slot_state(p, slot, state_set_age(slot_state(p, slot), v, p.clk));
}

// number of factors whose iters_since_relin is 0 (the loop of ba.py:96-99), one atomic per workgroup
__global__ __launch_bounds__(BLOCK) void k_count_relin(Params p, int *__restrict__ out)
{
    __shared__ int red[BLOCK / 64];
    const int slot = blockIdx.x * BLOCK + threadIdx.x;
    int cam, lmk;
    const bool hit = slot < p.T * WTILE && slot_info(p, slot, cam, lmk) && state_age(slot_state(p, slot), p.clk) == 0;
    const unsigned long long b = __ballot(hit);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) s += red[w];
        if (s) atomicAdd(out, s);
    }
}


// meas_fn / jac_fn of the reprojection factor at n free-standing points (reprojection.py:12-44): the unit the parity
// tests pin against fixture G1 -- the same `linearise` every sweep kernel inlines
__global__ __launch_bounds__(BLOCK) void k_eval_fn(Intrinsics K, int n, const double *__restrict__ x, double *__restrict__ h_out,
                                                   double *__restrict__ J_out, double *__restrict__ hproj_out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double x9[9], Jc[2][6], Jl[2][3], h[2], hp[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) x9[k] = x[(size_t)i * 9 + k];
    linearise(x9, K, Jc, Jl, h);
    project(x9, K, hp);
    if (h_out) { h_out[(size_t)i * 2] = h[0]; h_out[(size_t)i * 2 + 1] = h[1]; }
    if (hproj_out) { hproj_out[(size_t)i * 2] = hp[0]; hproj_out[(size_t)i * 2 + 1] = hp[1]; }
    if (J_out) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int k = 0; k < 6; ++k) J_out[(size_t)i * 18 + r * 9 + k] = Jc[r][k];
#pragma unroll
            for (int k = 0; k < 3; ++k) J_out[(size_t)i * 18 + r * 9 + 6 + k] = Jl[r][k];
        }
    }
}

// dense messages of a list of slots for the parity views (Factor.messages gbp.py:222): eta 6 | Lambda 21 packed | eta 3 | Lambda 6 packed
__global__ __launch_bounds__(BLOCK) void k_export_messages(Params p, const int *__restrict__ slots, int n, double *__restrict__ out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double eC[6], MC[21], eL[3], ML[6];
    dense_messages(p, slots[i], eC, MC, eL, ML);
    double *o = out + (size_t)i * 36;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = eC[k];
#pragma unroll
    for (int k = 0; k < 21; ++k) o[6 + k] = MC[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[27 + k] = eL[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) o[30 + k] = ML[k];
}

// Sigma = Lambda^-1 for the covariance view (VariableNode.Sigma gbp.py:192)
// mu of every variable, cameras then landmarks, dense: the viewer's per-frame read (vis/ba_vis.py:39-43, 111-114)
__global__ __launch_bounds__(BLOCK) void k_pack_means(Params p, double *__restrict__ out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < p.C * 6) out[i] = p.cbel[(size_t)(i / 6) * CAMREC + CAM_MU + i % 6];
    else if (i < p.C * 6 + p.L * 3) { const int j = i - p.C * 6; out[i] = p.lrec[(size_t)(j / 3) * LREC + LR_MU + j % 3]; }
}

__global__ __launch_bounds__(BLOCK) void k_covariances(Params p, double *__restrict__ cam_sig, double *__restrict__ lmk_sig)
{
    const int v = blockIdx.x * BLOCK + threadIdx.x;          // (the beliefs carry their covariances: gbp_math.hpp, covariance form)
    if (v < p.C) {
#pragma unroll
        for (int k = 0; k < 21; ++k) cam_sig[(size_t)v * 21 + k] = p.cbel[(size_t)v * CAMREC + CAM_COV + k];
    } else if (v < p.C + p.L) {
        const int l = v - p.C;
#pragma unroll
        for (int k = 0; k < 6; ++k) lmk_sig[(size_t)l * 6 + k] = p.lrec[(size_t)l * LREC + LR_COV + k];
    }
}


__global__ __launch_bounds__(BLOCK) void k_fill_iters(Params p, int n, int iters)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) set_slot_state(p, i, state_set_age(slot_state(p, i), min(max(iters, 0), ITERS_MAX), p.clk));
}


}  // namespace gbp
