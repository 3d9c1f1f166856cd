// gbp_math.hpp -- per-factor / per-variable fp64 device maths of the GBP bundle-adjustment sweep.
//
// Everything here runs in ONE LANE per factor (or per variable) with all state in VGPRs: every
// loop has compile-time bounds and is fully unrolled so no array is ever indexed at run time.
// fp64 throughout (fp32 storage or arithmetic cannot meet the 1e-4 gate).
//
// Reference maths restated (file:line of joeaortiz/gbp):
//   pin-hole projection + 2x9 Jacobian   gbp/factors/reprojection.py:12-44, utils/derivatives.py:36-50,
//                                        utils/lie_algebra.py:11-42, utils/transformations.py:5-7
//   linearisation  Lambda_f, eta_f       gbp/gbp.py:278-289
//   robust re-weighting                  gbp/gbp.py:296-332
//   relinearisation test                 gbp/gbp.py:70-80
//   factor->variable messages            gbp/gbp.py:334-373
//   belief = prior + sum(messages)       gbp/gbp.py:176-198
//
// Algebraic restructuring (same maths, fewer bytes and flops than the dense reference):
//   * Lambda_f = s J^T J is never materialised: with J = [Jc | Jl] (2x6 | 2x3) and rho = J x0 + z - h,
//       A = s Jc^T Jc, B = s Jc^T Jl, Cc = s Jl^T Jl, a = s Jc^T rho, c = s Jl^T rho.
//   * message to the camera   M' = A - B S^-1 B^T  with S = Cc + (Lambda_L - M_L)  becomes
//       M' = Jc^T (s I - s^2 Jl S^-1 Jl^T) Jc,   e* = s Jc^T (rho - Jl S^-1 g),  g = c + eta_L - e_L
//     and symmetrically for the landmark with the 6x6 T = A + (Lambda_C - M_C).
//   * S and T are symmetric positive definite (factor block + cavity >= prior), so the general LU
//     inverse of the reference is replaced by an unpivoted LDL^T on packed upper storage, and only
//     the forward substitution is needed: X^T S^-1 Y = (L^-1 X)^T D^-1 (L^-1 Y).
//   * symmetric matrices are stored packed (upper triangle, row-major): 6x6 -> 21, 3x3 -> 6;
//   * message precisions are stored as their 2x2 cores (M = J^T Q J): 3 doubles instead of 21 / 6 (rank2_update);
//   * message etas are stored as their 2-vectors of coefficients (e = J^T q): 2 doubles instead of 6 / 3.  The new eta is
//     s J^T (rho - ...) -- in the span of the Jacobian -- and damping (gbp.py:368) mixes it with the old one, which was made
//     with the same Jacobian unless the factor relinearised in this very sweep; then the damping is 0 (gbp.py:50-54: damping
//     returns num_undamped_iters sweeps after a relinearisation), so q' = (1-d) r + d q holds in both cases.  The one
//     configuration where it does not (num_undamped_iters = 0: damped in the relinearising sweep itself) carries the
//     out-of-span remainder in a dense side array (Params::xtra, general sweep only).
#pragma once
#include <hip/hip_runtime.h>

namespace gbp {

#define GBP_DEV __device__ __forceinline__

// 1/x for the pivots and depths of this path (normal, far from over/underflow): hardware seed (v_rcp_f64, ~26 bits)
// + two Newton steps = full double precision (<= 1 ulp) in 5 dependent instructions; the IEEE division sequence
// (v_div_scale / v_div_fmas / v_div_fixup) is twice as long and sits on every elimination's critical path.
GBP_DEV double rcp(double x)
{
#ifdef GBP_IEEE_DIV
    return 1.0 / x;
#else
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#endif
}

template <int N>
struct Sym {
    static constexpr int size = N * (N + 1) / 2;
    // packed index of (i, j), i <= j
    static constexpr __host__ __device__ int at(int i, int j) { return i * N - (i * (i - 1)) / 2 + (j - i); }
};

struct Intrinsics {
    double fx, fy, cx, cy;
};

// In-place LDL^T of a packed SPD matrix: on exit the diagonal holds D, the strict upper part holds
// U = L^T (unit upper), invd[k] = 1/D_k.
template <int N>
GBP_DEV void ldl_factor(double (&a)[Sym<N>::size], double (&invd)[N])
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double r = rcp(a[Sym<N>::at(k, k)]);
        invd[k] = r;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const double u = a[Sym<N>::at(k, i)] * r;
#pragma unroll
            for (int j = i; j < N; ++j) a[Sym<N>::at(i, j)] -= u * a[Sym<N>::at(k, j)];
            a[Sym<N>::at(k, i)] = u;
        }
    }
}

// b <- L^-1 b  (unit lower L = U^T from ldl_factor)
template <int N>
GBP_DEV void ldl_forward(const double (&a)[Sym<N>::size], double (&b)[N])
{
#pragma unroll
    for (int i = 1; i < N; ++i)
#pragma unroll
        for (int k = 0; k < i; ++k) b[i] -= a[Sym<N>::at(k, i)] * b[k];
}

// b <- L^-T b
template <int N>
GBP_DEV void ldl_backward(const double (&a)[Sym<N>::size], double (&b)[N])
{
#pragma unroll
    for (int i = N - 2; i >= 0; --i)
#pragma unroll
        for (int k = i + 1; k < N; ++k) b[i] -= a[Sym<N>::at(i, k)] * b[k];
}

// mu = Lambda^-1 eta for a packed SPD Lambda (VariableNode.update_belief gbp.py:192-193, and the
// belief means of gbp.py:74).  Lambda is consumed.
template <int N>
GBP_DEV void spd_solve(double (&lam)[Sym<N>::size], const double (&eta)[N], double (&mu)[N])
{
    double invd[N];
    ldl_factor<N>(lam, invd);
#pragma unroll
    for (int i = 0; i < N; ++i) mu[i] = eta[i];
    ldl_forward<N>(lam, mu);
#pragma unroll
    for (int i = 0; i < N; ++i) mu[i] *= invd[i];
    ldl_backward<N>(lam, mu);
}

// Sigma = Lambda^-1 (packed) from the LDL^T factors; used only by the covariance view.
template <int N>
GBP_DEV void spd_inverse(double (&lam)[Sym<N>::size], double (&sig)[Sym<N>::size])
{
    double invd[N];
    ldl_factor<N>(lam, invd);
#pragma unroll
    for (int c = 0; c < N; ++c) {
        double e[N];
#pragma unroll
        for (int i = 0; i < N; ++i) e[i] = (i == c) ? 1.0 : 0.0;
        ldl_forward<N>(lam, e);
#pragma unroll
        for (int i = 0; i < N; ++i) e[i] *= invd[i];
        ldl_backward<N>(lam, e);
#pragma unroll
        for (int i = 0; i <= c; ++i) sig[Sym<N>::at(i, c)] = e[i];
    }
}

// Rodrigues rotation of the axis-angle w (utils/lie_algebra.py:32-42) plus the three scalars the
// Jacobian of R(w) y needs.  Below 3*eps the reference returns R = I and its dR_wx_dw formula
// degenerates to -y^ (w w^T)/(w.w): mirrored through (c1, cI, cW) = (1/theta^2, 0, 0).
struct Rot {
    double r[3][3];
    double c1, cI, cW;   // (R^T - I) w^ + w w^T  ==  theta^2 * (c1 w w^T + cI I - cW w^)
};

// sin and cos of a non-negative angle with a small register footprint (the generic ocml sincos carries a
// Payne-Hanek path whose registers count against every lane even though axis-angle norms are O(1)):
// Cody-Waite reduction by pi/2 in three parts (exact products through fma; full accuracy for theta < ~1e9, degrading
// gracefully beyond), then the classic minimax kernels on [-pi/4, pi/4] (max error < 1 ulp).
// A double literal that is not an inline constant needs a register pair; left to itself the compiler parks every one of
// them in VGPRs outside the persistent loop of the fused sweep (20 registers held -- and spilled -- through the whole tile).
// Born in SGPRs at the point of use they cost two s_mov each and no vector register (a VALU op takes one scalar operand).
GBP_DEV double sconst(double c)
{
    asm volatile("" : "+s"(c));
    return c;
}

GBP_DEV void sincos_theta(double x, double &sn, double &cs)
{
    const double n = rint(x * sconst(6.36619772367581382433e-01));           // 2/pi
    double r = fma(n, sconst(-1.5707963267948966e+00), x);
    r = fma(n, sconst(-6.123233995736766e-17), r);
    r = fma(n, sconst(1.4973849048591698e-33), r);
    const double z = r * r;
    // sin(r) = r + r^3 (S1 + z (S2 + ...)),  cos(r) = 1 - z/2 + z^2 (C1 + z (C2 + ...))
    double ps = fma(z, sconst(1.58969099521155010221e-10), sconst(-2.50507602534068634195e-08));
    ps = fma(z, ps, sconst(2.75573137070700676789e-06));
    ps = fma(z, ps, sconst(-1.98412698298579493134e-04));
    ps = fma(z, ps, sconst(8.33333333332248946124e-03));
    ps = fma(z, ps, sconst(-1.66666666666666324348e-01));
    double pc = fma(z, sconst(-1.13596475577881948265e-11), sconst(2.08757232129817482790e-09));
    pc = fma(z, pc, sconst(-2.75573143513906633035e-07));
    pc = fma(z, pc, sconst(2.48015872894767294178e-05));
    pc = fma(z, pc, sconst(-1.38888888888741095749e-03));
    pc = fma(z, pc, sconst(4.16666666666666019037e-02));
    const double s0 = fma(r * z, ps, r);
    const double c0 = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)n & 3;
    const double ss = (q & 1) ? c0 : s0, cc = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -ss : ss;
    cs = ((q + 1) & 2) ? -cc : cc;
}

GBP_DEV Rot rodrigues(double w0, double w1, double w2)
{
    Rot o;
    const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
    const double th = sqrt(th2);
    if (th < 3.0 * 2.220446049250313e-16) {
        o.r[0][0] = 1.0; o.r[0][1] = 0.0; o.r[0][2] = 0.0;
        o.r[1][0] = 0.0; o.r[1][1] = 1.0; o.r[1][2] = 0.0;
        o.r[2][0] = 0.0; o.r[2][1] = 0.0; o.r[2][2] = 1.0;
        o.c1 = 1.0 / th2; o.cI = 0.0; o.cW = 0.0;
        return o;
    }
    double sn, cs;
#ifdef GBP_OCML_SINCOS
    sincos(th, &sn, &cs);
#else
    sincos_theta(th, sn, cs);
#endif
    const double ith2 = rcp(th2);
    const double a = sn * rcp(th);
    const double b = (1.0 - cs) * ith2;
    // R = I + a w^ + b w^ w^,  w^ w^ = w w^T - theta^2 I  (diagonal written without cancellation)
    o.r[0][0] = 1.0 - b * (w1 * w1 + w2 * w2);
    o.r[1][1] = 1.0 - b * (w0 * w0 + w2 * w2);
    o.r[2][2] = 1.0 - b * (w0 * w0 + w1 * w1);
    const double b01 = b * w0 * w1, b02 = b * w0 * w2, b12 = b * w1 * w2;
    o.r[0][1] = b01 - a * w2; o.r[1][0] = b01 + a * w2;
    o.r[0][2] = b02 + a * w1; o.r[2][0] = b02 - a * w1;
    o.r[1][2] = b12 - a * w0; o.r[2][1] = b12 + a * w0;
    o.c1 = (1.0 - a) * ith2; o.cI = a; o.cW = b;
    return o;
}

// h(x) = proj(K (R(w) y + t))   reprojection.py:12-24
GBP_DEV void project(const double (&x)[9], const Intrinsics &K, double (&h)[2])
{
    const Rot R = rodrigues(x[3], x[4], x[5]);
    const double p0 = R.r[0][0] * x[6] + R.r[0][1] * x[7] + R.r[0][2] * x[8] + x[0];
    const double p1 = R.r[1][0] * x[6] + R.r[1][1] * x[7] + R.r[1][2] * x[8] + x[1];
    const double p2 = R.r[2][0] * x[6] + R.r[2][1] * x[7] + R.r[2][2] * x[8] + x[2];
    const double iz = 1.0 / p2;
    h[0] = (K.fx * p0 + K.cx * p2) * iz;
    h[1] = (K.fy * p1 + K.cy * p2) * iz;
}

// h(x) and J(x) = [J_p K | J_p K dR_wx_dw(w, y) | J_p K R]   reprojection.py:27-44
GBP_DEV void linearise(const double (&x)[9], const Intrinsics &K, double (&Jc)[2][6], double (&Jl)[2][3],
                       double (&h)[2])
{
    const double w0 = x[3], w1 = x[4], w2 = x[5], y0 = x[6], y1 = x[7], y2 = x[8];
    const Rot R = rodrigues(w0, w1, w2);
    const double p0 = R.r[0][0] * y0 + R.r[0][1] * y1 + R.r[0][2] * y2 + x[0];
    const double p1 = R.r[1][0] * y0 + R.r[1][1] * y1 + R.r[1][2] * y2 + x[1];
    const double p2 = R.r[2][0] * y0 + R.r[2][1] * y1 + R.r[2][2] * y2 + x[2];
    const double iz = rcp(p2);
    h[0] = (K.fx * p0 + K.cx * p2) * iz;
    h[1] = (K.fy * p1 + K.cy * p2) * iz;
    // J_p K = [[fx/Z, 0, cx/Z - X/Z^2], [0, fy/Z, cy/Z - Y/Z^2]];  cx/Z - X/Z^2 == -fx p0 / Z^2
    const double a0 = K.fx * iz, a2 = -a0 * p0 * iz;
    const double b1 = K.fy * iz, b2 = -b1 * p1 * iz;
    Jc[0][0] = a0;  Jc[0][1] = 0.0; Jc[0][2] = a2;
    Jc[1][0] = 0.0; Jc[1][1] = b1;  Jc[1][2] = b2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        Jl[0][j] = a0 * R.r[0][j] + a2 * R.r[2][j];
        Jl[1][j] = b1 * R.r[1][j] + b2 * R.r[2][j];
    }
    // J_p K dR_wx_dw = -(J_p K R) y^ (c1 w w^T + cI I - cW w^)      derivatives.py:36-45
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double q0 = Jl[r][1] * y2 - Jl[r][2] * y1;
        const double q1 = Jl[r][2] * y0 - Jl[r][0] * y2;
        const double q2 = Jl[r][0] * y1 - Jl[r][1] * y0;
        const double qw = (q0 * w0 + q1 * w1 + q2 * w2) * R.c1;
        Jc[r][3] = -(qw * w0 + R.cI * q0 - R.cW * (q1 * w2 - q2 * w1));
        Jc[r][4] = -(qw * w1 + R.cI * q1 - R.cW * (q2 * w0 - q0 * w2));
        Jc[r][5] = -(qw * w2 + R.cI * q2 - R.cW * (q0 * w1 - q1 * w0));
    }
}

// Factor.robustify_loss: adaptive noise variance from the residual AT THE LINEARISATION POINT
// (gbp.py:309-328).  loss: 1 huber, 2 constant ("m^2" without sigma^2 is the reference's, gbp.py:324).
GBP_DEV double robust_variance(int loss, double sigma2, double nstds, double r0, double r1, bool &flag)
{
    const double m = sqrt(r0 * r0 + r1 * r1) / sqrt(sigma2);
    flag = m > nstds;
    if (!flag) return sigma2;
    if (loss == 1) return sigma2 * (m * m) / (2.0 * (nstds * m - 0.5 * (nstds * nstds)));
    return m * m;
}

// Linearised factor in the compact form the messages need.
struct Lin {
    double Jc[2][6], Jl[2][3], rho[2];   // J = [Jc | Jl], rho = J x0 + z - h(x0)
    double s, d;                         // 1 / adaptive variance, eta damping
};

// Message to the LANDMARK: eliminate the camera block (6x6).   Factor.compute_messages, v = 1  gbp.py:340-368
//   cavity of the camera: cetaC = eta_C - e_C, clamC = Lambda_C - M_C (belief minus this factor's OLD message)
//   T = s Jc^T Jc + clamC,  u = s Jc^T rho + cetaC          (assembled by the caller: factor_core, gbp_kernels.hpp)
//   M_L' = Jl^T (sI - s^2 Jc T^-1 Jc^T) Jl,  e_L' = (1-d) s Jl^T (rho - Jc T^-1 u) + d e_L
// The forward substitutions of Jc^T and u ride along with the LDL^T elimination (augmented columns) and the
// 2x2 quadratic forms are accumulated pivot by pivot, so nothing but the shrinking trailing block stays live.
//   qLold / qLnew: coefficients of the message eta in the rows of Jl (e_L = Jl^T q_L), eLnew = the dense new eta
GBP_DEV void message_to_landmark(const Lin &L, double (&u)[6], double (&clamC)[21],
                                 const double (&qLold)[2], double (&qLnew)[2], double (&eLnew)[3], double (&MLnew)[6],
                                 double (&Vcore)[3])
{
    const double s = L.s;
    double y0[6], y1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { y0[i] = L.Jc[0][i]; y1[i] = L.Jc[1][i]; }
    double H00 = 0.0, H01 = 0.0, H11 = 0.0, k0 = 0.0, k1 = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double r = rcp(clamC[Sym<6>::at(k, k)]);
        const double t0 = y0[k] * r, t1 = y1[k] * r;
        H00 += t0 * y0[k]; H01 += t0 * y1[k]; H11 += t1 * y1[k];
        k0 += t0 * u[k]; k1 += t1 * u[k];
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double m = clamC[Sym<6>::at(k, i)] * r;
#pragma unroll
            for (int j = i; j < 6; ++j) clamC[Sym<6>::at(i, j)] -= m * clamC[Sym<6>::at(k, j)];
            y0[i] -= m * y0[k]; y1[i] -= m * y1[k]; u[i] -= m * u[k];
        }
    }
    const double V00 = s - s * s * H00, V01 = -(s * s) * H01, V11 = s - s * s * H11;
    Vcore[0] = V00; Vcore[1] = V01; Vcore[2] = V11;
    const double r0 = s * (L.rho[0] - k0), r1 = s * (L.rho[1] - k1);
    double VJ0[3], VJ1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        VJ0[j] = V00 * L.Jl[0][j] + V01 * L.Jl[1][j];
        VJ1[j] = V01 * L.Jl[0][j] + V11 * L.Jl[1][j];
    }
    qLnew[0] = (1.0 - L.d) * r0 + L.d * qLold[0];
    qLnew[1] = (1.0 - L.d) * r1 + L.d * qLold[1];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        eLnew[i] = L.Jl[0][i] * qLnew[0] + L.Jl[1][i] * qLnew[1];
#pragma unroll
        for (int j = i; j < 3; ++j) MLnew[Sym<3>::at(i, j)] = L.Jl[0][i] * VJ0[j] + L.Jl[1][i] * VJ1[j];
    }
}

// Message to the CAMERA: eliminate the landmark block (3x3).   Factor.compute_messages, v = 0  gbp.py:340-368
//   cavity of the landmark: cetaL = eta_L - e_L, clamL = Lambda_L - M_L (OLD landmark message)
//   S = s Jl^T Jl + clamL,  g = s Jl^T rho + cetaL          (assembled by the caller)
//   M_C' = Jc^T (sI - s^2 Jl S^-1 Jl^T) Jc,  e_C' = (1-d) s Jc^T (rho - Jl S^-1 g) + d e_C = Jc^T q_C'   (qC in: old, out: new)
GBP_DEV void message_to_camera(const Lin &L, double (&g)[3], double (&clamL)[6],
                               double (&qC)[2], double (&eCnew)[6], double (&MCnew)[21], double (&Wcore)[3])
{
    const double s = L.s;
    double y0[3], y1[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { y0[i] = L.Jl[0][i]; y1[i] = L.Jl[1][i]; }
    double G00 = 0.0, G01 = 0.0, G11 = 0.0, k0 = 0.0, k1 = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double r = rcp(clamL[Sym<3>::at(k, k)]);
        const double t0 = y0[k] * r, t1 = y1[k] * r;
        G00 += t0 * y0[k]; G01 += t0 * y1[k]; G11 += t1 * y1[k];
        k0 += t0 * g[k]; k1 += t1 * g[k];
#pragma unroll
        for (int i = k + 1; i < 3; ++i) {
            const double m = clamL[Sym<3>::at(k, i)] * r;
#pragma unroll
            for (int j = i; j < 3; ++j) clamL[Sym<3>::at(i, j)] -= m * clamL[Sym<3>::at(k, j)];
            y0[i] -= m * y0[k]; y1[i] -= m * y1[k]; g[i] -= m * g[k];
        }
    }
    const double W00 = s - s * s * G00, W01 = -(s * s) * G01, W11 = s - s * s * G11;
    Wcore[0] = W00; Wcore[1] = W01; Wcore[2] = W11;
    const double r0 = s * (L.rho[0] - k0), r1 = s * (L.rho[1] - k1);
    double WJ0[6], WJ1[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        WJ0[j] = W00 * L.Jc[0][j] + W01 * L.Jc[1][j];
        WJ1[j] = W01 * L.Jc[0][j] + W11 * L.Jc[1][j];
    }
    qC[0] = (1.0 - L.d) * r0 + L.d * qC[0];
    qC[1] = (1.0 - L.d) * r1 + L.d * qC[1];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        eCnew[i] = L.Jc[0][i] * qC[0] + L.Jc[1][i] * qC[1];
#pragma unroll
        for (int j = i; j < 6; ++j) MCnew[Sym<6>::at(i, j)] = L.Jc[0][i] * WJ0[j] + L.Jc[1][i] * WJ1[j];
    }
}

// A message precision never leaves the span of its Jacobian block: M = J^T Q J with a symmetric 2x2 core Q (W for the
// camera message, V for the landmark message; Lambda is never damped, gbp.py:368).  Only Q (3 doubles) is stored per
// message instead of the packed 21 / 6; the dense matrix is rebuilt from the Jacobian at the linearisation point the
// message was computed with.   T += sign * J^T Q J   on packed upper storage, J given by its two rows.
template <int N>
GBP_DEV void rank2_update(double (&T)[Sym<N>::size], const double (&j0)[N], const double (&j1)[N], const double (&Q)[3],
                          double sign)
{
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double a = sign * (Q[0] * j0[j] + Q[1] * j1[j]);
        const double b = sign * (Q[1] * j0[j] + Q[2] * j1[j]);
#pragma unroll
        for (int i = 0; i <= j; ++i) T[Sym<N>::at(i, j)] += j0[i] * a + j1[i] * b;
    }
}

// max over all 81 signed entries of Lambda_f = s J^T J (np.max(factor.factor.lam), gbp_ba.py:31)
GBP_DEV double factor_lambda_max(const double (&Jc)[2][6], const double (&Jl)[2][3], double s)
{
    double J0[9], J1[9];
#pragma unroll
    for (int i = 0; i < 6; ++i) { J0[i] = Jc[0][i]; J1[i] = Jc[1][i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { J0[6 + i] = Jl[0][i]; J1[6 + i] = Jl[1][i]; }
    double m = s * (J0[0] * J0[0] + J1[0] * J1[0]);
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = i; j < 9; ++j) m = fmax(m, s * (J0[i] * J0[j] + J1[i] * J1[j]));
    return m;
}

}  // namespace gbp
