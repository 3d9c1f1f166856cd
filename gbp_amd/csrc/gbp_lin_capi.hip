// gbp_lin_capi.hip -- linear pairwise GBP on gfx950 behind include/gbp_lin.h.
//
// The reference's generic path (gbp/gbp.py FactorGraph with nonlinear_factors=False, ndim_posegraph.py) for graphs
// of two-variable factors over d-dimensional variables.  Two kernels per sweep, like the general BA sweep:
//   k_lin_factor<D>   one lane per factor: both outgoing messages from the OLD incoming ones
//                     (Factor.compute_messages gbp.py:334-373): cavity of the other variable folded into its block,
//                     that block eliminated by an unpivoted LDL^T (SPD: factor block + prior-backed cavity);
//   k_lin_belief<D>   64/(d+P) variables per wave, one lane per belief entry: prior + messages in adj_factors order from
//                     the variable-major copy of the messages (one contiguous run per variable), mean by a d x d solve
//                     (VariableNode.update_belief gbp.py:176-198).
// Layout: everything factor-indexed is SoA [row][F] (coalesced across lanes); the new messages are ALSO written in CSR
// edge order (vmsg, through an LDS transpose) for the belief stage; beliefs are records [N][d + d(d+1)/2 + d]
// (eta | packed Lambda | mu) gathered per factor.  Lambda_f is constant: packed upper 2d x 2d, read every sweep.
// HBM-bound like the BA sweep; d <= 6 keeps a factor's working set in registers (one wave per SIMD for d = 6).
#include "../../include/gbp_ba.h"
#include "../../include/gbp_lin.h"
#include "gbp_math.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

namespace gbp {
int set_error(int code, const char *fmt, ...);      // gbp_capi.hip: thread-local message behind gbp_last_error()

struct LinParams {
    int N, F;
    double damping;
    const int *va, *vb;          // [F]
    const double *feta, *flam;   // [2D][F], [D(2D+1)][F] packed upper
    const double *fconst;        // [F]
    double *msg_a, *msg_b;       // [D + D(D+1)/2][F] each: eta rows then packed Lambda rows
    double *bel;                 // [N][D + P + D]
    const double *prior;         // [N][D + P]
    const int *vptr, *vadj;      // CSR: variable -> (factor << 1 | side), ascending factor id
    double *vmsg;                // [2F][D + P]: the same messages in VARIABLE-major (CSR edge) order, for the belief stage
    const int *epos_a, *epos_b;  // [F]: CSR edge index of (factor, side)
};

template <int D> struct LinDims {
    static constexpr int P = D * (D + 1) / 2;       // packed d x d
    static constexpr int P2 = D * (2 * D + 1);      // packed 2d x 2d
    static constexpr int REC = D + P + D;           // belief record
};

// Message to the KEPT variable from  [ A_kk  A_kn ; A_nk  A_nn ] , with the eliminated block S = A_nn + cavity already
// formed (consumed):  Lambda = A_kk - A_kn S^-1 A_nk,  eta = e_k - A_kn S^-1 e_n   (gbp.py:353-367).
// akn(i, j) = A[k_i][n_j] is supplied by the caller as a dense D x D array.
template <int D>
GBP_DEV void lin_schur(const double (&akk)[Sym<D>::size], const double (&akn)[D][D], double (&S)[Sym<D>::size],
                       const double (&ek)[D], double (&en)[D], double (&lam)[Sym<D>::size], double (&eta)[D])
{
    double invd[D];
    ldl_factor<D>(S, invd);
    ldl_forward<D>(S, en);                            // L^-1 e_n
    double y[D][D];                                   // y[i] = L^-1 (A_nk column i) = L^-1 akn[i][:]
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) y[i][j] = akn[i][j];
        ldl_forward<D>(S, y[i]);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double e = ek[i];
#pragma unroll
        for (int k = 0; k < D; ++k) e -= y[i][k] * invd[k] * en[k];
        eta[i] = e;
#pragma unroll
        for (int j = i; j < D; ++j) {
            double v = akk[Sym<D>::at(i, j)];
#pragma unroll
            for (int k = 0; k < D; ++k) v -= y[i][k] * invd[k] * y[j][k];
            lam[Sym<D>::at(i, j)] = v;
        }
    }
}

template <int D>
__global__ __launch_bounds__(64) void k_lin_factor(LinParams p)
{
    constexpr int P = LinDims<D>::P, REC = LinDims<D>::REC, R = D + P;
    __shared__ double tr[64 * R];                           // transposes a wave's messages for the variable-major copy
    __shared__ int tp[64];
    const int lane = threadIdx.x;
    const int nlive = min(64, p.F - (int)blockIdx.x * 64);  // factors of this wave (> 0 by the launch grid)
    const int f = blockIdx.x * 64 + min(lane, nlive - 1);   // lanes past the end redo the last factor and store nothing
    const bool live = lane < nlive;
    const size_t F = (size_t)p.F;
    // the belief stage reads messages in CSR edge order: stage this wave's 64 records in LDS and write them out as
    // contiguous (D+P)-double runs, 64/(D+P) records per store instruction (a lane-per-record store would touch 64 lines)
    auto stage_out = [&](const double (&eta)[D], const double (&lam)[P], const int *epos) {
#pragma unroll
        for (int k = 0; k < D; ++k) tr[lane * R + k] = eta[k];
#pragma unroll
        for (int k = 0; k < P; ++k) tr[lane * R + D + k] = lam[k];
        tp[lane] = epos[f];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave per block
        constexpr int G = 64 / R;
        const int g = lane / R, k = lane - g * R;
        if (g < G) {
            if (nlive == 64) {                              // full wave: fixed trip count, LDS reads batched ahead of the stores
#pragma unroll
                for (int j0 = 0; j0 < 64; j0 += G) {
                    const int j = j0 + g;
                    if (j < 64) p.vmsg[(size_t)tp[j] * R + k] = tr[j * R + k];
                }
            } else {
                for (int j = g; j < nlive; j += G) p.vmsg[(size_t)tp[j] * R + k] = tr[j * R + k];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    const double *ra = p.bel + (size_t)p.va[f] * REC, *rb = p.bel + (size_t)p.vb[f] * REC;
    // cavities: belief minus this factor's old message (gbp.py:341-350)
    double cea[D], cla[P], ceb[D], clb[P], oea[D], oeb[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { oea[k] = p.msg_a[k * F + f]; oeb[k] = p.msg_b[k * F + f]; cea[k] = ra[k] - oea[k]; ceb[k] = rb[k] - oeb[k]; }
#pragma unroll
    for (int k = 0; k < P; ++k) { cla[k] = ra[D + k] - p.msg_a[(D + k) * F + f]; clb[k] = rb[D + k] - p.msg_b[(D + k) * F + f]; }
    // the factor: packed upper 2D x 2D = [ A_aa A_ab ; . A_bb ]
    double aaa[P], abb[P], aab[D][D], aba[D][D], fa[D], fb[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { fa[k] = p.feta[k * F + f]; fb[k] = p.feta[(D + k) * F + f]; }
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = i; j < D; ++j) {
            aaa[Sym<D>::at(i, j)] = p.flam[(size_t)Sym<2 * D>::at(i, j) * F + f];
            abb[Sym<D>::at(i, j)] = p.flam[(size_t)Sym<2 * D>::at(D + i, D + j) * F + f];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) { const double v = p.flam[(size_t)Sym<2 * D>::at(i, D + j) * F + f]; aab[i][j] = v; aba[j][i] = v; }
    }
    const double d = p.damping;
    double lam[P], eta[D], S[P], en[D];
    // to a: eliminate b
#pragma unroll
    for (int k = 0; k < P; ++k) S[k] = abb[k] + clb[k];
#pragma unroll
    for (int k = 0; k < D; ++k) en[k] = fb[k] + ceb[k];
    lin_schur<D>(aaa, aab, S, fa, en, lam, eta);
#pragma unroll
    for (int k = 0; k < D; ++k) eta[k] = (1.0 - d) * eta[k] + d * oea[k];                   // gbp.py:368
    if (live) {
#pragma unroll
        for (int k = 0; k < D; ++k) p.msg_a[k * F + f] = eta[k];
#pragma unroll
        for (int k = 0; k < P; ++k) p.msg_a[(D + k) * F + f] = lam[k];
    }
    stage_out(eta, lam, p.epos_a);
    // to b: eliminate a (from the OLD message of b: both are committed together, gbp.py:371-373 -- cea / cla were
    // formed before the store above)
#pragma unroll
    for (int k = 0; k < P; ++k) S[k] = aaa[k] + cla[k];
#pragma unroll
    for (int k = 0; k < D; ++k) en[k] = fa[k] + cea[k];
    lin_schur<D>(abb, aba, S, fb, en, lam, eta);
#pragma unroll
    for (int k = 0; k < D; ++k) eta[k] = (1.0 - d) * eta[k] + d * oeb[k];
    if (live) {
#pragma unroll
        for (int k = 0; k < D; ++k) p.msg_b[k * F + f] = eta[k];
#pragma unroll
        for (int k = 0; k < P; ++k) p.msg_b[(D + k) * F + f] = lam[k];
    }
    stage_out(eta, lam, p.epos_b);
}

// 64/(D+P) variables per wave, one lane per (variable, belief entry): the variable's messages are one contiguous run of
// vmsg (CSR edge order = adj_factors order), so the wave streams whole lines; the d x d solve is done by the entry-0 lane.
template <int D>
__global__ __launch_bounds__(64) void k_lin_belief(LinParams p)
{
    constexpr int P = LinDims<D>::P, REC = LinDims<D>::REC, R = D + P, VPW = 64 / R;
    __shared__ double sh[VPW * R];
    const int lane = threadIdx.x, j = lane / R, k = lane - j * R;
    const int v = blockIdx.x * VPW + j;
    const bool on = j < VPW && v < p.N;
    if (on) {
        double acc = p.prior[(size_t)v * R + k];
        const int e1 = p.vptr[v + 1];
        int e = p.vptr[v];
        for (; e + 3 < e1; e += 4) {                              // four loads in flight, added in adj_factors order (gbp.py:182-188)
            const double m0 = p.vmsg[(size_t)e * R + k], m1 = p.vmsg[(size_t)(e + 1) * R + k], m2 = p.vmsg[(size_t)(e + 2) * R + k],
                         m3 = p.vmsg[(size_t)(e + 3) * R + k];
            acc += m0; acc += m1; acc += m2; acc += m3;
        }
        for (; e < e1; ++e) acc += p.vmsg[(size_t)e * R + k];
        p.bel[(size_t)v * REC + k] = acc;
        sh[j * R + k] = acc;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // one wave per block
    if (on && k == 0) {
        double eta[D], lam[P], mu[D];
#pragma unroll
        for (int i = 0; i < D; ++i) eta[i] = sh[j * R + i];
#pragma unroll
        for (int i = 0; i < P; ++i) lam[i] = sh[j * R + D + i];
        spd_solve<D>(lam, eta, mu);                               // gbp.py:192-193
#pragma unroll
        for (int i = 0; i < D; ++i) p.bel[(size_t)v * REC + R + i] = mu[i];
    }
}

// sum over factors of 0.5 mu^T Lambda_f mu - eta_f^T mu + const = 0.5 |h(mu) - z|^2 / sigma^2 for linear h (gbp.py:36-44, 261-265)
template <int D>
__global__ __launch_bounds__(256) void k_lin_energy(LinParams p, double *out)
{
    constexpr int P = LinDims<D>::P, REC = LinDims<D>::REC;
    __shared__ double red[256 / 64];
    const int f = blockIdx.x * 256 + threadIdx.x;
    double e = 0.0;
    if (f < p.F) {
        const size_t F = (size_t)p.F;
        double x[2 * D];
        const double *ra = p.bel + (size_t)p.va[f] * REC + D + P, *rb = p.bel + (size_t)p.vb[f] * REC + D + P;
#pragma unroll
        for (int k = 0; k < D; ++k) { x[k] = ra[k]; x[D + k] = rb[k]; }
        e = p.fconst ? p.fconst[f] : 0.0;
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) {
            e -= p.feta[i * F + f] * x[i];
#pragma unroll
            for (int j = i; j < 2 * D; ++j) e += (i == j ? 0.5 : 1.0) * p.flam[(size_t)Sym<2 * D>::at(i, j) * F + f] * x[i] * x[j];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e += __shfl_down(e, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace gbp

using namespace gbp;

struct gbp_lin {
    LinParams p{};
    int D = 0, device = 0;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    double *d_red = nullptr;
    int red_blocks = 0;
    bool has_beliefs = false;
};

#define LHIPCHK(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess)                                                                              \
            return set_error(e__ == hipErrorOutOfMemory ? GBP_ENOMEM : GBP_EHIP, "%s failed: %s (%s:%d)",   \
                             #expr, hipGetErrorString(e__), __FILE__, __LINE__);                            \
    } while (0)
#define LCHK(expr) do { int rc__ = (expr); if (rc__ != GBP_OK) return rc__; } while (0)
#define LENTER(h)                                                                        \
    do {                                                                                 \
        if (!(h)) return set_error(GBP_EINVAL, "NULL handle");                           \
        LHIPCHK(hipSetDevice((h)->device));                                              \
    } while (0)

template <typename T>
static int lin_upload(gbp_lin *h, T **out, const std::vector<T> &v)
{
    void *q = nullptr;
    LHIPCHK(hipMalloc(&q, std::max<size_t>(v.size(), 1) * sizeof(T)));
    h->allocs.push_back(q);
    if (!v.empty()) LHIPCHK(hipMemcpyAsync(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    LHIPCHK(hipStreamSynchronize(h->stream));            // `v` may be a temporary of the caller
    *out = static_cast<T *>(q);
    return GBP_OK;
}

template <typename K>
static void lin_dispatch(int D, K &&k)
{
    switch (D) {
    case 1: k(std::integral_constant<int, 1>{}); break;
    case 2: k(std::integral_constant<int, 2>{}); break;
    case 3: k(std::integral_constant<int, 3>{}); break;
    case 4: k(std::integral_constant<int, 4>{}); break;
    case 5: k(std::integral_constant<int, 5>{}); break;
    default: k(std::integral_constant<int, 6>{}); break;
    }
}

static int lin_beliefs(gbp_lin *h)
{
    if (h->p.N) lin_dispatch(h->D, [&](auto d) {
        constexpr int DD = decltype(d)::value, VPW = 64 / (DD + DD * (DD + 1) / 2);
        hipLaunchKernelGGL((k_lin_belief<DD>), dim3((h->p.N + VPW - 1) / VPW), dim3(64), 0, h->stream, h->p);
    });
    LHIPCHK(hipGetLastError());
    h->has_beliefs = true;
    return GBP_OK;
}

static int lin_create_impl(gbp_lin *h, const gbp_lin_desc_t *d)
{
    const int N = d->n_vars, F = d->n_factors, D = d->dofs;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(GBP_ENODEV, "no HIP device visible: libgbp_hip.so has no CPU path");
    if (d->device < 0 || d->device >= ndev) return set_error(GBP_EINVAL, "device %d out of range (%d visible)", d->device, ndev);
    h->device = d->device;
    LHIPCHK(hipSetDevice(h->device));
    hipDeviceProp_t prop;
    LHIPCHK(hipGetDeviceProperties(&prop, h->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(GBP_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only", h->device, prop.gcnArchName);
    LHIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->D = D;
    const int P = D * (D + 1) / 2, P2 = D * (2 * D + 1), D2 = 2 * D;
    LinParams &p = h->p;
    p.N = N; p.F = F; p.damping = d->eta_damping;

    // factor data -> SoA, packed upper; adjacency CSR in ascending factor id (= append order, ndim_posegraph.py:86-88)
    std::vector<int32_t> va(d->var_a, d->var_a + F), vb(d->var_b, d->var_b + F), vptr((size_t)N + 1, 0), vadj((size_t)2 * F);
    std::vector<double> feta((size_t)D2 * F), flam((size_t)P2 * F), fconst((size_t)F, 0.0);
    for (int f = 0; f < F; ++f) {
        if (va[f] < 0 || va[f] >= N || vb[f] < 0 || vb[f] >= N || va[f] == vb[f])
            return set_error(GBP_EINVAL, "factor %d joins variables (%d, %d): need two different ids in [0, %d)", f, va[f], vb[f], N);
        ++vptr[va[f] + 1]; ++vptr[vb[f] + 1];
        for (int i = 0; i < D2; ++i) {
            feta[(size_t)i * F + f] = d->factor_eta[(size_t)f * D2 + i];
            for (int j = i; j < D2; ++j) {
                const int at = i * D2 - (i * (i - 1)) / 2 + (j - i);
                flam[(size_t)at * F + f] = d->factor_lam[((size_t)f * D2 + i) * D2 + j];
            }
        }
        if (d->factor_const) fconst[f] = d->factor_const[f];
    }
    for (int v = 0; v < N; ++v) vptr[v + 1] += vptr[v];
    std::vector<int32_t> epos_a((size_t)F), epos_b((size_t)F);
    {
        std::vector<int32_t> fill(vptr.begin(), vptr.end() - 1);
        for (int f = 0; f < F; ++f) {
            epos_a[f] = fill[va[f]]; vadj[fill[va[f]]++] = f << 1;
            epos_b[f] = fill[vb[f]]; vadj[fill[vb[f]]++] = (f << 1) | 1;
        }
    }
    std::vector<double> prior((size_t)N * (D + P)), zeros_msg((size_t)(D + P) * F, 0.0), zeros_bel((size_t)N * (D + P + D), 0.0);
    for (int v = 0; v < N; ++v) {
        for (int i = 0; i < D; ++i) {
            prior[(size_t)v * (D + P) + i] = d->prior_eta[(size_t)v * D + i];
            for (int j = i; j < D; ++j) prior[(size_t)v * (D + P) + D + i * D - (i * (i - 1)) / 2 + (j - i)] = d->prior_lam[((size_t)v * D + i) * D + j];
        }
    }
    int *dva, *dvb, *dvptr, *dvadj;
    double *dfeta, *dflam, *dfconst, *dprior;
    LCHK(lin_upload(h, &dva, va)); LCHK(lin_upload(h, &dvb, vb)); LCHK(lin_upload(h, &dvptr, vptr)); LCHK(lin_upload(h, &dvadj, vadj));
    LCHK(lin_upload(h, &dfeta, feta)); LCHK(lin_upload(h, &dflam, flam)); LCHK(lin_upload(h, &dfconst, fconst)); LCHK(lin_upload(h, &dprior, prior));
    LCHK(lin_upload(h, &p.msg_a, zeros_msg)); LCHK(lin_upload(h, &p.msg_b, zeros_msg)); LCHK(lin_upload(h, &p.bel, zeros_bel));
    {
        std::vector<double> zeros_v((size_t)2 * F * (D + P), 0.0);
        int *dea, *deb;
        LCHK(lin_upload(h, &p.vmsg, zeros_v)); LCHK(lin_upload(h, &dea, epos_a)); LCHK(lin_upload(h, &deb, epos_b));
        p.epos_a = dea; p.epos_b = deb;
    }
    p.va = dva; p.vb = dvb; p.vptr = dvptr; p.vadj = dvadj; p.feta = dfeta; p.flam = dflam; p.fconst = dfconst; p.prior = dprior;
    h->red_blocks = std::max(1, (F + 255) / 256);
    std::vector<double> zr((size_t)h->red_blocks, 0.0);
    LCHK(lin_upload(h, &h->d_red, zr));
    return GBP_OK;
}

extern "C" {

int gbp_lin_create(gbp_lin_t **out, const gbp_lin_desc_t *d)
{
    if (!out || !d) return set_error(GBP_EINVAL, "NULL argument");
    *out = nullptr;
    if (d->n_vars < 0 || d->n_factors < 0) return set_error(GBP_EINVAL, "negative size");
    if (d->dofs < 1 || d->dofs > GBP_LIN_MAX_DOFS) return set_error(GBP_EINVAL, "dofs %d not in 1..%d", d->dofs, GBP_LIN_MAX_DOFS);
    if ((d->n_factors && (!d->var_a || !d->var_b || !d->factor_eta || !d->factor_lam)) || (d->n_vars && (!d->prior_eta || !d->prior_lam)))
        return set_error(GBP_EINVAL, "NULL array in the descriptor");
    gbp_lin *h = new (std::nothrow) gbp_lin;
    if (!h) return set_error(GBP_ENOMEM, "out of host memory");
    const int rc = lin_create_impl(h, d);
    if (rc != GBP_OK) { gbp_lin_destroy(h); return rc; }
    *out = h;
    return GBP_OK;
}

void gbp_lin_destroy(gbp_lin_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *q : h->allocs) (void)hipFree(q);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int gbp_lin_sync(gbp_lin_t *h)
{
    LENTER(h);
    LHIPCHK(hipStreamSynchronize(h->stream));
    return GBP_OK;
}

int gbp_lin_update_beliefs(gbp_lin_t *h)
{
    LENTER(h);
    return lin_beliefs(h);
}

int gbp_lin_iterate(gbp_lin_t *h, int32_t n_iters)
{
    LENTER(h);
    if (n_iters < 0) return set_error(GBP_EINVAL, "negative iteration count");
    if (!h->has_beliefs) return set_error(GBP_ESTATE, "call gbp_lin_update_beliefs first (ndim_posegraph.py:90)");
    for (int it = 0; it < n_iters; ++it) {
        if (h->p.F) lin_dispatch(h->D, [&](auto d) {
            hipLaunchKernelGGL((k_lin_factor<decltype(d)::value>), dim3((h->p.F + 63) / 64), dim3(64), 0, h->stream, h->p);
        });
        LCHK(lin_beliefs(h));
    }
    LHIPCHK(hipGetLastError());
    return GBP_OK;
}

int gbp_lin_energy(gbp_lin_t *h, double *out)
{
    LENTER(h);
    if (!out) return set_error(GBP_EINVAL, "out is NULL");
    if (!h->has_beliefs) return set_error(GBP_ESTATE, "beliefs have not been computed yet");
    double e = 0.0;
    if (h->p.F) {
        lin_dispatch(h->D, [&](auto d) {
            hipLaunchKernelGGL((k_lin_energy<decltype(d)::value>), dim3(h->red_blocks), dim3(256), 0, h->stream, h->p, h->d_red);
        });
        LHIPCHK(hipGetLastError());
        std::vector<double> r((size_t)h->red_blocks);
        LHIPCHK(hipMemcpyAsync(r.data(), h->d_red, r.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        LHIPCHK(hipStreamSynchronize(h->stream));
        for (double v : r) e += v;
    }
    *out = e;
    return GBP_OK;
}

static int lin_download(gbp_lin *h, std::vector<double> &dst, const double *src, size_t n)
{
    dst.resize(n);
    if (n) LHIPCHK(hipMemcpyAsync(dst.data(), src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    LHIPCHK(hipStreamSynchronize(h->stream));
    return GBP_OK;
}

static void unpack_sym(int D, const double *packed, size_t stride, double *dense)
{
    for (int i = 0; i < D; ++i)
        for (int j = i; j < D; ++j) {
            const double v = packed[(size_t)(i * D - (i * (i - 1)) / 2 + (j - i)) * stride];
            dense[i * D + j] = v; dense[j * D + i] = v;
        }
}

int gbp_lin_get_beliefs(gbp_lin_t *h, double *eta, double *lam)
{
    LENTER(h);
    const int D = h->D, P = D * (D + 1) / 2, REC = D + P + D;
    std::vector<double> b;
    LCHK(lin_download(h, b, h->p.bel, (size_t)h->p.N * REC));
    for (int v = 0; v < h->p.N; ++v) {
        if (eta) for (int k = 0; k < D; ++k) eta[(size_t)v * D + k] = b[(size_t)v * REC + k];
        if (lam) unpack_sym(D, &b[(size_t)v * REC + D], 1, lam + (size_t)v * D * D);
    }
    return GBP_OK;
}

int gbp_lin_get_means(gbp_lin_t *h, double *mu)
{
    LENTER(h);
    if (!mu) return set_error(GBP_EINVAL, "mu is NULL");
    const int D = h->D, P = D * (D + 1) / 2, REC = D + P + D;
    std::vector<double> b;
    LCHK(lin_download(h, b, h->p.bel, (size_t)h->p.N * REC));
    for (int v = 0; v < h->p.N; ++v) for (int k = 0; k < D; ++k) mu[(size_t)v * D + k] = b[(size_t)v * REC + D + P + k];
    return GBP_OK;
}

int gbp_lin_get_messages(gbp_lin_t *h, double *eta_a, double *lam_a, double *eta_b, double *lam_b)
{
    LENTER(h);
    const int D = h->D, P = D * (D + 1) / 2;
    const size_t F = (size_t)h->p.F;
    for (int side = 0; side < 2; ++side) {
        double *eta = side ? eta_b : eta_a, *lam = side ? lam_b : lam_a;
        if (!eta && !lam) continue;
        std::vector<double> m;
        LCHK(lin_download(h, m, side ? h->p.msg_b : h->p.msg_a, (size_t)(D + P) * F));
        for (size_t f = 0; f < F; ++f) {
            if (eta) for (int k = 0; k < D; ++k) eta[f * D + k] = m[(size_t)k * F + f];
            if (lam) unpack_sym(D, &m[(size_t)D * F + f], F, lam + f * D * D);
        }
    }
    return GBP_OK;
}

}  // extern "C"
