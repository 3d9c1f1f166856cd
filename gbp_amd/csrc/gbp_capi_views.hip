// gbp_capi_views.hip -- libgbp_hip.so, the state views of include/gbp_ba.h (reference order, dense): beliefs, means, covariances, priors,
// messages, factors, relinearisation state and its setters, the streaming means export for a viewer, and gbp_ba_eval_fn.  Each view
// gathers on the device and moves only the requested range (gbp_view_kernels.hpp).
#include "gbp_handle.hpp"
#include "gbp_view_kernels.hpp"

extern "C" {

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


}  // extern "C"
