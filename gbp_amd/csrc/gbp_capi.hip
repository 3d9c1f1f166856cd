// gbp_capi.hip -- host side of libgbp_hip.so: graph lay-out, launches and the C ABI of include/gbp_ba.h.
//
// No CPU compute path lives here: every sweep, belief update and diagnostic is a HIP kernel.  Host
// code only (a) permutes the caller's arrays between the reference's factor order and the internal
// landmark-major layout, (b) packs/unpacks symmetric matrices for the views, (c) does the one-off
// per-variable max of generate_priors_var (gbp_ba.py:27-31) over factor maxima computed on device.
#include "../../include/gbp_ba.h"
#include "gbp_kernels.hpp"
#include "gbp_fused.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
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

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(e__ == hipErrorOutOfMemory ? GBP_ENOMEM : GBP_EHIP, "%s failed: %s (%s:%d)",   \
                        #expr, hipGetErrorString(e__), __FILE__, __LINE__);                            \
    } while (0)

#define CHK(expr) do { int rc__ = (expr); if (rc__ != GBP_OK) return rc__; } while (0)

static inline int round_up(int n, int m) { return (n + m - 1) / m * m; }

struct gbp_ba {
    Params p{};
    int device = 0;
    int flags = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // host-side order maps
    std::vector<int32_t> int2ref, ref2int;       // internal <-> reference factor ids
    std::vector<int32_t> ref_cam, ref_lmk;       // per reference factor
    std::vector<int32_t> h_lptr, h_cptr;         // CSR offsets (internal / reference order)
    // device scratch
    double *d_partial = nullptr;                 // C*27 camera partial sums (single-GPU path)
    double *d_red = nullptr;                     // per-block residual partials
    double *d_tmp = nullptr; size_t tmp_bytes = 0;
    int *d_ids = nullptr; size_t ids_cap = 0;
    std::vector<void *> allocs;
    bool has_beliefs = false;
    // fused path
    FusedPlan fused;
    // timing of the dominant kernel
    bool timing = false;
    std::vector<hipEvent_t> ev;                  // pairs
    size_t ev_used = 0;
    const char *dominant = "k_factor";
};

template <typename T>
static int dev_alloc(gbp_ba *h, T **out, size_t n, bool zero = true)
{
    void *ptr = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHK(hipMalloc(&ptr, bytes));
    h->allocs.push_back(ptr);
    if (zero) HIPCHK(hipMemsetAsync(ptr, 0, bytes, h->stream));
    *out = static_cast<T *>(ptr);
    return GBP_OK;
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

// ------------------------------------------------------------------------------ launches --

static int time_begin(gbp_ba *h)
{
    if (!h->timing) return GBP_OK;
    if (h->ev_used + 2 > h->ev.size()) {
        for (int i = 0; i < 2; ++i) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->ev.push_back(e); }
    }
    HIPCHK(hipEventRecord(h->ev[h->ev_used], h->stream));
    return GBP_OK;
}

static int time_end(gbp_ba *h)
{
    if (!h->timing) return GBP_OK;
    HIPCHK(hipEventRecord(h->ev[h->ev_used + 1], h->stream));
    h->ev_used += 2;
    return GBP_OK;
}

static int launch_factor_stage(gbp_ba *h, int robustify, int local_relin)
{
    Params p = h->p;
    p.robustify = robustify; p.local_relin = local_relin;
    if (!p.F) return GBP_OK;
    CHK(time_begin(h));
    switch (p.loss) {
    case GBP_LOSS_NONE: hipLaunchKernelGGL(k_factor<0>, dim3(grid_for(p.F)), dim3(BLOCK), 0, h->stream, p); break;
    case GBP_LOSS_HUBER: hipLaunchKernelGGL(k_factor<1>, dim3(grid_for(p.F)), dim3(BLOCK), 0, h->stream, p); break;
    default: hipLaunchKernelGGL(k_factor<2>, dim3(grid_for(p.F)), dim3(BLOCK), 0, h->stream, p); break;
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

static int launch_cam_finish(gbp_ba *h, const double *gathered, int n_parts, size_t stride)
{
    if (!h->p.C) return GBP_OK;
    hipLaunchKernelGGL(k_cam_finish, dim3((h->p.C + 63) / 64), dim3(64), 0, h->stream, h->p, gathered, n_parts, stride);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

// one synchronous_iteration's device work up to (and including) this rank's camera partial sums
static int sweep_begin(gbp_ba *h, int with_messages, int robustify, int local_relin, double *partial)
{
    if (with_messages && h->fused.enabled) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (h->timing) {
            if (h->ev_used + 2 > h->ev.size())
                for (int i = 0; i < 2; ++i) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->ev.push_back(e); }
            e0 = h->ev[h->ev_used]; e1 = h->ev[h->ev_used + 1];
            h->ev_used += 2;
        }
        int rc = fused_launch(h->fused, h->p, robustify, local_relin, partial, h->stream, e0, e1);
        if (rc != 0) return fail(GBP_EHIP, "fused sweep launch failed: %s", hipGetErrorString((hipError_t)rc));
        return GBP_OK;
    }
    if (with_messages) CHK(launch_factor_stage(h, robustify, local_relin));
    CHK(launch_lmk_beliefs(h));
    CHK(launch_cam_partial(h, partial));
    return GBP_OK;
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
    if (h->d_ids) (void)hipFree(h->d_ids);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    fused_destroy(h->fused);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

static int create_impl(gbp_ba *h, const gbp_ba_desc_t *d)
{
    const int C = d->n_cams, L = d->n_lmks, F = d->n_factors;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GBP_ENODEV, "no HIP device visible: libgbp_hip.so has no CPU path");
    if (d->device < 0 || d->device >= ndev) return fail(GBP_EINVAL, "device %d out of range (%d visible)", d->device, ndev);
    h->device = d->device;
    HIPCHK(hipSetDevice(h->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, h->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GBP_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only", h->device, prop.gcnArchName);
    HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    h->flags = d->flags;

    Params &p = h->p;
    p.F = F; p.L = L; p.C = C;
    p.Fp = round_up(std::max(F, 1), BLOCK); p.Lp = round_up(std::max(L, 1), BLOCK);
    p.K = Intrinsics{d->K[0], d->K[1], d->K[2], d->K[3]};
    p.sigma2 = d->gauss_noise_std * d->gauss_noise_std;
    p.nstds = d->nstds; p.beta = d->beta; p.eta_damping = d->eta_damping;
    p.num_undamped = d->num_undamped_iters; p.min_linear = d->min_linear_iters; p.loss = d->loss;
    p.robustify = 0; p.local_relin = 1;

    // reference order: camera-major, stable in file order (gbp_ba.py:128-130)
    h->h_cptr.assign((size_t)C + 1, 0);
    for (int i = 0; i < F; ++i) {
        if (d->cam_idx[i] < 0 || d->cam_idx[i] >= C || d->lmk_idx[i] < 0 || d->lmk_idx[i] >= L)
            return fail(GBP_EINVAL, "observation %d references camera %d / landmark %d outside [0,%d) / [0,%d)",
                        i, d->cam_idx[i], d->lmk_idx[i], C, L);
        h->h_cptr[(size_t)d->cam_idx[i] + 1]++;
    }
    for (int c = 0; c < C; ++c) h->h_cptr[c + 1] += h->h_cptr[c];
    std::vector<int32_t> ref_file((size_t)F);
    {
        std::vector<int32_t> cur(h->h_cptr.begin(), h->h_cptr.end() - 1);
        for (int i = 0; i < F; ++i) ref_file[(size_t)cur[d->cam_idx[i]]++] = i;
    }
    h->ref_cam.resize(F); h->ref_lmk.resize(F);
    for (int r = 0; r < F; ++r) { h->ref_cam[r] = d->cam_idx[ref_file[r]]; h->ref_lmk[r] = d->lmk_idx[ref_file[r]]; }
    // internal order: landmark-major, stable in reference id (= VariableNode.adj_factors order, gbp_ba.py:139)
    h->h_lptr.assign((size_t)L + 1, 0);
    for (int r = 0; r < F; ++r) h->h_lptr[(size_t)h->ref_lmk[r] + 1]++;
    for (int l = 0; l < L; ++l) h->h_lptr[l + 1] += h->h_lptr[l];
    h->int2ref.resize(F); h->ref2int.resize(F);
    {
        std::vector<int32_t> cur(h->h_lptr.begin(), h->h_lptr.end() - 1);
        for (int r = 0; r < F; ++r) { int i = cur[h->ref_lmk[r]]++; h->int2ref[i] = r; h->ref2int[r] = i; }
    }

    const size_t Fp = p.Fp, Lp = p.Lp;
    std::vector<double> x0(9 * Fp, 0.0), z(2 * Fp, 0.0);
    std::vector<int32_t> fcam(Fp, 0), flmk(Fp, 0), state(Fp, 1 << STATE_SHIFT);     // iters_since_relin = 1  gbp.py:249
    for (int i = 0; i < F; ++i) {
        const int r = h->int2ref[i], c = h->ref_cam[r], l = h->ref_lmk[r], fi = ref_file[r];
        for (int k = 0; k < 6; ++k) x0[k * Fp + i] = d->cam_means[(size_t)c * 6 + k];     // linpoint = concat(cam.mu, lmk.mu) gbp_ba.py:136
        for (int k = 0; k < 3; ++k) x0[(6 + k) * Fp + i] = d->lmk_means[(size_t)l * 3 + k];
        z[i] = d->meas[(size_t)fi * 2]; z[Fp + i] = d->meas[(size_t)fi * 2 + 1];
        fcam[i] = c; flmk[i] = l;
    }
    CHK(dev_alloc(h, &p.x0, 9 * Fp)); CHK(dev_alloc(h, &p.z, 2 * Fp));
    CHK(dev_alloc(h, &p.mc, 27 * Fp)); CHK(dev_alloc(h, &p.ml, 9 * Fp));
    CHK(dev_alloc(h, &p.fcam, Fp)); CHK(dev_alloc(h, &p.flmk, Fp)); CHK(dev_alloc(h, &p.state, Fp));
    if (p.loss != GBP_LOSS_NONE) {
        CHK(dev_alloc(h, &p.avar, Fp));
        std::vector<double> av(Fp, p.sigma2);                                             // gbp.py:242
        CHK(upload(h, p.avar, av));
    }
    CHK(upload(h, p.x0, x0)); CHK(upload(h, p.z, z));
    CHK(upload(h, p.fcam, fcam)); CHK(upload(h, p.flmk, flmk));

    CHK(dev_alloc(h, &p.lbel, 9 * Lp)); CHK(dev_alloc(h, &p.lmu, 3 * Lp)); CHK(dev_alloc(h, &p.lprior, 9 * Lp));
    CHK(dev_alloc(h, &p.cbel, (size_t)std::max(C, 1) * CAMREC)); CHK(dev_alloc(h, &p.cprior, (size_t)std::max(C, 1) * 27));
    {
        std::vector<double> lmu(3 * Lp, 0.0), cb((size_t)std::max(C, 1) * CAMREC, 0.0);
        for (int l = 0; l < L; ++l) for (int k = 0; k < 3; ++k) lmu[k * Lp + l] = d->lmk_means[(size_t)l * 3 + k];   // node.mu = init gbp_ba.py:123
        for (int c = 0; c < C; ++c) for (int k = 0; k < 6; ++k) cb[(size_t)c * CAMREC + CAM_MU + k] = d->cam_means[(size_t)c * 6 + k];
        CHK(upload(h, p.lmu, lmu)); CHK(upload(h, p.cbel, cb));
    }
    int *lptr = nullptr, *cptr = nullptr, *cadj = nullptr;
    CHK(dev_alloc(h, &lptr, (size_t)L + 1)); CHK(dev_alloc(h, &cptr, (size_t)C + 1)); CHK(dev_alloc(h, &cadj, (size_t)std::max(F, 1)));
    CHK(upload(h, lptr, h->h_lptr)); CHK(upload(h, cptr, h->h_cptr)); CHK(upload(h, cadj, h->ref2int));
    p.lptr = lptr; p.cptr = cptr; p.cadj = cadj;

    CHK(dev_alloc(h, &h->d_partial, (size_t)std::max(C, 1) * 27));
    CHK(dev_alloc(h, &h->d_red, 2 * (size_t)grid_for(Fp)));

    if (!(h->flags & GBP_FLAG_NO_FUSED)) {
        int rc = fused_plan(h->fused, p, h->h_lptr, fcam, state, h->stream, prop.multiProcessorCount);
        if (rc < 0) return fail(GBP_EHIP, "building the fused sweep plan failed (%d)", rc);
        if (h->fused.enabled) h->dominant = "k_sweep_fused";
    }
    CHK(upload(h, p.state, state));                      // after the plan: it embeds the per-tile ranks
    HIPCHK(hipStreamSynchronize(h->stream));
    return GBP_OK;
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

int gbp_ba_sync(gbp_ba_t *h)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    return GBP_OK;
}

// ------------------------------------------------------------------------------- priors ---

int gbp_ba_factor_lambda_max(gbp_ba_t *h, double *cam_max, double *lmk_max)
{
    ENTER(h);
    const Params &p = h->p;
    CHK(ensure_tmp(h, sizeof(double) * (size_t)p.Fp));
    if (p.F) hipLaunchKernelGGL(k_factor_lambda_max, dim3(grid_for(p.F)), dim3(BLOCK), 0, h->stream, p, h->d_tmp);
    HIPCHK(hipGetLastError());
    std::vector<double> fm;
    CHK(download(h, fm, h->d_tmp, (size_t)p.F));
    // max_factor_lam = 0.; max over adjacent factors (gbp_ba.py:27-31)
    if (lmk_max)
        for (int l = 0; l < p.L; ++l) {
            double m = 0.0;
            for (int i = h->h_lptr[l]; i < h->h_lptr[l + 1]; ++i) m = std::max(m, fm[i]);
            lmk_max[l] = m;
        }
    if (cam_max)
        for (int c = 0; c < p.C; ++c) {
            double m = 0.0;
            for (int r = h->h_cptr[c]; r < h->h_cptr[c + 1]; ++r) m = std::max(m, fm[h->ref2int[r]]);
            cam_max[c] = m;
        }
    return GBP_OK;
}

int gbp_ba_set_prior_scalars(gbp_ba_t *h, const double *cam_lambda, const double *lmk_lambda)
{
    ENTER(h);
    const Params &p = h->p;
    if (!cam_lambda || !lmk_lambda) return fail(GBP_EINVAL, "null argument");
    std::vector<double> cb, lmu;
    CHK(download(h, cb, p.cbel, (size_t)std::max(p.C, 1) * CAMREC));
    CHK(download(h, lmu, p.lmu, 3 * (size_t)p.Lp));
    std::vector<double> cp((size_t)std::max(p.C, 1) * 27, 0.0), lp(9 * (size_t)p.Lp, 0.0);
    for (int c = 0; c < p.C; ++c) {                 // lam_prior = eye * l; eta = lam_prior @ mu  (gbp_ba.py:32-34)
        for (int k = 0; k < 6; ++k) {
            cp[(size_t)c * 27 + k] = cam_lambda[c] * cb[(size_t)c * CAMREC + CAM_MU + k];
            cp[(size_t)c * 27 + 6 + Sym<6>::at(k, k)] = cam_lambda[c];
        }
    }
    for (int l = 0; l < p.L; ++l) {
        for (int k = 0; k < 3; ++k) {
            lp[(size_t)k * p.Lp + l] = lmk_lambda[l] * lmu[(size_t)k * p.Lp + l];
            lp[(size_t)(3 + Sym<3>::at(k, k)) * p.Lp + l] = lmk_lambda[l];
        }
    }
    CHK(upload(h, p.cprior, cp)); CHK(upload(h, p.lprior, lp));
    return GBP_OK;
}

int gbp_ba_generate_priors(gbp_ba_t *h, double weaker_factor)
{
    ENTER(h);
    if (!(weaker_factor != 0.0)) return fail(GBP_EINVAL, "weaker_factor must be non-zero");
    std::vector<double> cm((size_t)h->p.C), lm((size_t)h->p.L);
    CHK(gbp_ba_factor_lambda_max(h, cm.data(), lm.data()));
    const double w2 = weaker_factor * weaker_factor;
    for (double &v : cm) v = v / w2;
    for (double &v : lm) v = v / w2;
    return gbp_ba_set_prior_scalars(h, cm.data(), lm.data());
}

int gbp_ba_set_priors(gbp_ba_t *h, const double *cam_eta, const double *cam_lam, const double *lmk_eta, const double *lmk_lam)
{
    ENTER(h);
    const Params &p = h->p;
    if (!cam_eta || !cam_lam || !lmk_eta || !lmk_lam) return fail(GBP_EINVAL, "null argument");
    std::vector<double> cp((size_t)std::max(p.C, 1) * 27, 0.0), lp(9 * (size_t)p.Lp, 0.0);
    for (int c = 0; c < p.C; ++c) {
        for (int k = 0; k < 6; ++k) cp[(size_t)c * 27 + k] = cam_eta[(size_t)c * 6 + k];
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j)
                cp[(size_t)c * 27 + 6 + Sym<6>::at(i, j)] = 0.5 * (cam_lam[(size_t)c * 36 + i * 6 + j] + cam_lam[(size_t)c * 36 + j * 6 + i]);
    }
    for (int l = 0; l < p.L; ++l) {
        for (int k = 0; k < 3; ++k) lp[(size_t)k * p.Lp + l] = lmk_eta[(size_t)l * 3 + k];
        for (int i = 0; i < 3; ++i)
            for (int j = i; j < 3; ++j)
                lp[(size_t)(3 + Sym<3>::at(i, j)) * p.Lp + l] = 0.5 * (lmk_lam[(size_t)l * 9 + i * 3 + j] + lmk_lam[(size_t)l * 9 + j * 3 + i]);
    }
    CHK(upload(h, p.cprior, cp)); CHK(upload(h, p.lprior, lp));
    return GBP_OK;
}

int gbp_ba_weaken_priors(gbp_ba_t *h, double factor)
{
    ENTER(h);
    const Params &p = h->p;
    const size_t nc = (size_t)p.C * 27, nl = 9 * (size_t)p.Lp;
    if (nc) hipLaunchKernelGGL(k_scale, dim3(grid_for(nc)), dim3(BLOCK), 0, h->stream, p.cprior, nc, factor);
    if (p.L) hipLaunchKernelGGL(k_scale, dim3(grid_for(nl)), dim3(BLOCK), 0, h->stream, p.lprior, nl, factor);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

// -------------------------------------------------------------------------------- sweep ---

int gbp_ba_update_beliefs(gbp_ba_t *h)
{
    ENTER(h);
    CHK(sweep_begin(h, 0, 0, 0, h->d_partial));
    CHK(launch_cam_finish(h, h->d_partial, 1, 0));
    h->has_beliefs = true;
    return GBP_OK;
}

int gbp_ba_iterate(gbp_ba_t *h, int32_t n_iters, int32_t robustify, int32_t local_relin)
{
    ENTER(h);
    if (n_iters < 0) return fail(GBP_EINVAL, "n_iters < 0");
    for (int it = 0; it < n_iters; ++it) {
        CHK(sweep_begin(h, 1, robustify, local_relin, h->d_partial));
        CHK(launch_cam_finish(h, h->d_partial, 1, 0));
    }
    h->has_beliefs = true;
    return GBP_OK;
}

int gbp_ba_shard_begin(gbp_ba_t *h, int32_t with_messages, int32_t robustify, int32_t local_relin, double *partial_dev)
{
    ENTER(h);
    if (!partial_dev) return fail(GBP_EINVAL, "null partial buffer");
    return sweep_begin(h, with_messages, robustify, local_relin, partial_dev);
}

int gbp_ba_shard_end(gbp_ba_t *h, const double *gathered_dev, int32_t n_ranks)
{
    ENTER(h);
    if (!gathered_dev || n_ranks < 1) return fail(GBP_EINVAL, "bad gathered buffer / rank count");
    CHK(launch_cam_finish(h, gathered_dev, n_ranks, (size_t)h->p.C * 27));
    h->has_beliefs = true;
    return GBP_OK;
}

// --------------------------------------------------------------------------- diagnostics ---

int gbp_ba_residual_sums(gbp_ba_t *h, double out[2])
{
    ENTER(h);
    if (!out) return fail(GBP_EINVAL, "null argument");
    const Params &p = h->p;
    out[0] = out[1] = 0.0;
    if (!p.F) return GBP_OK;
    const int nb = grid_for(p.F);
    hipLaunchKernelGGL(k_residual, dim3(nb), dim3(BLOCK), 0, h->stream, p, h->d_red);
    HIPCHK(hipGetLastError());
    std::vector<double> part;
    CHK(download(h, part, h->d_red, 2 * (size_t)nb));
    for (int b = 0; b < nb; ++b) { out[0] += part[2 * b]; out[1] += part[2 * b + 1]; }
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

static int get_var_info(gbp_ba *h, const double *d_cam, int cam_stride, int cam_off, const double *d_lmk,
                        double *cam_eta, double *cam_lam, double *lmk_eta, double *lmk_lam)
{
    const Params &p = h->p;
    if (cam_eta || cam_lam) {
        std::vector<double> cb;
        CHK(download(h, cb, d_cam, (size_t)std::max(p.C, 1) * cam_stride));
        for (int c = 0; c < p.C; ++c) {
            if (cam_eta) for (int k = 0; k < 6; ++k) cam_eta[(size_t)c * 6 + k] = cb[(size_t)c * cam_stride + cam_off + k];
            if (cam_lam) unpack6(&cb[(size_t)c * cam_stride + cam_off + 6], cam_lam + (size_t)c * 36);
        }
    }
    if (lmk_eta || lmk_lam) {
        std::vector<double> lb;
        CHK(download(h, lb, d_lmk, 9 * (size_t)p.Lp));
        for (int l = 0; l < p.L; ++l) {
            if (lmk_eta) for (int k = 0; k < 3; ++k) lmk_eta[(size_t)l * 3 + k] = lb[(size_t)k * p.Lp + l];
            if (lmk_lam) {
                double pk[6];
                for (int k = 0; k < 6; ++k) pk[k] = lb[(size_t)(3 + k) * p.Lp + l];
                unpack3(pk, lmk_lam + (size_t)l * 9);
            }
        }
    }
    return GBP_OK;
}

int gbp_ba_get_beliefs(gbp_ba_t *h, double *cam_eta, double *cam_lam, double *lmk_eta, double *lmk_lam)
{
    ENTER(h);
    return get_var_info(h, h->p.cbel, CAMREC, CAM_ETA, h->p.lbel, cam_eta, cam_lam, lmk_eta, lmk_lam);
}

int gbp_ba_get_priors(gbp_ba_t *h, double *cam_eta, double *cam_lam, double *lmk_eta, double *lmk_lam)
{
    ENTER(h);
    return get_var_info(h, h->p.cprior, 27, 0, h->p.lprior, cam_eta, cam_lam, lmk_eta, lmk_lam);
}

int gbp_ba_get_means(gbp_ba_t *h, double *cam_mu, double *lmk_mu)
{
    ENTER(h);
    const Params &p = h->p;
    if (cam_mu) {
        std::vector<double> cb;
        CHK(download(h, cb, p.cbel, (size_t)std::max(p.C, 1) * CAMREC));
        for (int c = 0; c < p.C; ++c) for (int k = 0; k < 6; ++k) cam_mu[(size_t)c * 6 + k] = cb[(size_t)c * CAMREC + CAM_MU + k];
    }
    if (lmk_mu) {
        std::vector<double> lm;
        CHK(download(h, lm, p.lmu, 3 * (size_t)p.Lp));
        for (int l = 0; l < p.L; ++l) for (int k = 0; k < 3; ++k) lmk_mu[(size_t)l * 3 + k] = lm[(size_t)k * p.Lp + l];
    }
    return GBP_OK;
}

int gbp_ba_get_covariances(gbp_ba_t *h, double *cam_sigma, double *lmk_sigma)
{
    ENTER(h);
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
    const size_t Fp = p.Fp;
    if (cam_eta || cam_lam) {
        std::vector<double> mc;
        CHK(download(h, mc, p.mc, 27 * Fp));
        for (int q = 0; q < n; ++q) {
            const int i = h->ref2int[f0 + q];
            if (cam_eta) for (int k = 0; k < 6; ++k) cam_eta[(size_t)q * 6 + k] = mc[k * Fp + i];
            if (cam_lam) { double pk[21]; for (int k = 0; k < 21; ++k) pk[k] = mc[(6 + k) * Fp + i]; unpack6(pk, cam_lam + (size_t)q * 36); }
        }
    }
    if (lmk_eta || lmk_lam) {
        std::vector<double> ml;
        CHK(download(h, ml, p.ml, 9 * Fp));
        for (int q = 0; q < n; ++q) {
            const int i = h->ref2int[f0 + q];
            if (lmk_eta) for (int k = 0; k < 3; ++k) lmk_eta[(size_t)q * 3 + k] = ml[k * Fp + i];
            if (lmk_lam) { double pk[6]; for (int k = 0; k < 6; ++k) pk[k] = ml[(3 + k) * Fp + i]; unpack3(pk, lmk_lam + (size_t)q * 9); }
        }
    }
    return GBP_OK;
}

int gbp_ba_get_factors(gbp_ba_t *h, int32_t f0, int32_t n, double *eta, double *lam, double *linpoint, int32_t *cam, int32_t *lmk, double *meas)
{
    ENTER(h);
    CHK(check_range(h, f0, n));
    const Params &p = h->p;
    const size_t Fp = p.Fp;
    if (cam) for (int q = 0; q < n; ++q) cam[q] = h->ref_cam[f0 + q];
    if (lmk) for (int q = 0; q < n; ++q) lmk[q] = h->ref_lmk[f0 + q];
    if (linpoint) {
        std::vector<double> x0;
        CHK(download(h, x0, p.x0, 9 * Fp));
        for (int q = 0; q < n; ++q) for (int k = 0; k < 9; ++k) linpoint[(size_t)q * 9 + k] = x0[k * Fp + h->ref2int[f0 + q]];
    }
    if (meas) {
        std::vector<double> z;
        CHK(download(h, z, p.z, 2 * Fp));
        for (int q = 0; q < n; ++q) { meas[(size_t)q * 2] = z[h->ref2int[f0 + q]]; meas[(size_t)q * 2 + 1] = z[Fp + h->ref2int[f0 + q]]; }
    }
    if ((eta || lam) && n) {
        if ((size_t)n > h->ids_cap) {
            if (h->d_ids) HIPCHK(hipFree(h->d_ids));
            h->d_ids = nullptr; h->ids_cap = 0;
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->d_ids), sizeof(int) * (size_t)n));
            h->ids_cap = n;
        }
        HIPCHK(hipMemcpyAsync(h->d_ids, h->ref2int.data() + f0, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
        CHK(ensure_tmp(h, sizeof(double) * 90 * (size_t)n));
        hipLaunchKernelGGL(k_export_factors, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, p, h->d_ids, n, h->d_tmp, h->d_tmp + 9 * (size_t)n);
        HIPCHK(hipGetLastError());
        if (eta) HIPCHK(hipMemcpyAsync(eta, h->d_tmp, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        if (lam) HIPCHK(hipMemcpyAsync(lam, h->d_tmp + 9 * (size_t)n, sizeof(double) * 81 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return GBP_OK;
}

int gbp_ba_get_relin_state(gbp_ba_t *h, int32_t *iters, double *eta_damping, double *adaptive_var, uint8_t *robust_flag)
{
    ENTER(h);
    const Params &p = h->p;
    std::vector<int32_t> st;
    CHK(download(h, st, p.state, (size_t)p.Fp));
    std::vector<double> av;
    if (adaptive_var && p.loss != GBP_LOSS_NONE) CHK(download(h, av, p.avar, (size_t)p.Fp));
    for (int r = 0; r < p.F; ++r) {
        const int s = st[h->ref2int[r]];
        if (iters) iters[r] = s >> STATE_SHIFT;
        if (eta_damping) eta_damping[r] = (s & 1) ? p.eta_damping : 0.0;
        if (robust_flag) robust_flag[r] = (uint8_t)((s >> 1) & 1);
        if (adaptive_var) adaptive_var[r] = p.loss != GBP_LOSS_NONE ? av[h->ref2int[r]] : p.sigma2;
    }
    return GBP_OK;
}

int gbp_ba_set_iters_since_relin(gbp_ba_t *h, const int32_t *iters)
{
    ENTER(h);
    if (!iters) return fail(GBP_EINVAL, "null argument");
    const Params &p = h->p;
    std::vector<int32_t> st;
    CHK(download(h, st, p.state, (size_t)p.Fp));
    for (int r = 0; r < p.F; ++r) {
        int32_t &s = st[h->ref2int[r]];
        s = (int32_t)(((uint32_t)iters[r] << STATE_SHIFT) | ((uint32_t)s & ((1u << STATE_SHIFT) - 1u)));
    }
    CHK(upload(h, p.state, st));
    return GBP_OK;
}

int gbp_ba_fill_iters_since_relin(gbp_ba_t *h, int32_t value)
{
    ENTER(h);
    if (h->p.F) hipLaunchKernelGGL(k_fill_iters, dim3(grid_for(h->p.F)), dim3(BLOCK), 0, h->stream, h->p.state, h->p.F, value);
    HIPCHK(hipGetLastError());
    return GBP_OK;
}

// ------------------------------------------------------------------------ instrumentation ---

int gbp_ba_set_kernel_timing(gbp_ba_t *h, int32_t enable)
{
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->timing = enable != 0;
    h->ev_used = 0;
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

int gbp_ba_info(gbp_ba_t *h, int32_t *fused_path, int32_t *n_tiles, int32_t *n_blocks)
{
    if (!h) return fail(GBP_EINVAL, "null handle");
    if (fused_path) *fused_path = h->fused.enabled ? 1 : 0;
    if (n_tiles) *n_tiles = h->fused.n_tiles;
    if (n_blocks) *n_blocks = h->fused.n_blocks;
    return GBP_OK;
}

}  // extern "C"
