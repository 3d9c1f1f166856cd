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
// Algebraic restructuring (same maths as the dense reference, fewer bytes and flops):
//   * Lambda_f = s J^T J is never materialised: J = [Jc | Jl] (2x6 | 2x3), rho = J x0 + z - h, s = 1 / adaptive variance.
//   * A message never leaves the span of its factor's Jacobian and is stored that way: precision M = J^T W J with a symmetric
//     2x2 core (3 doubles instead of 21 / 6), eta e = J^T q with a coefficient pair (2 doubles instead of 6 / 3).  The new eta
//     is s J^T (rho - ...) -- in the span -- and damping (gbp.py:368) mixes it with the old one, which was made with the same
//     Jacobian unless the factor relinearised in this very sweep; then the damping is 0 (gbp.py:50-54), so
//     q' = (1-d) r + d q holds in both cases.  The one configuration where it does not (damped in the relinearising sweep
//     itself) carries the out-of-span remainder in a dense side array (Params::xtra, general sweep only).
//   * COVARIANCE FORM (round 4).  The reference inverts, per FACTOR and per message, the factor's own block plus the cavity
//     of the variable it eliminates (gbp.py:340-368: a 6x6 inverse for the message to the landmark, a 3x3 for the message to
//     the camera).  With P = Lambda^-1 of the variable's BELIEF -- one inverse per VARIABLE, made where the belief is made --
//     and the old message in core form, the matrix to invert is a rank-2 update of the belief,
//         T = Lambda - J^T W J + s J^T J = Lambda + J^T Q J,   Q = s I - W,
//     and everything the message needs is 2x2 algebra on G = J P J^T (Woodbury / push-through identities):
//         J T^-1 J^T            = G (I + Q G)^-1
//         core of the message   = s I - s^2 J T^-1 J^T = s (I - W G) (I + Q G)^-1          (no cancellation left in it)
//         coefficients of eta   = s (rho - J T^-1 u)   = s (I + G Q)^-1 [ rho - J mu + G (q - W rho) ],   u = eta + J^T (s rho - q)
//     (J, W, q, P, mu of the ELIMINATED variable; the result is the message to the other one).  Nothing of the belief but
//     its mean and covariance is read by a factor.  A factor that relinearises takes its old message out first, with the old
//     Jacobian -- P' = P + P J^T (I - W G)^-1 W J P, mu' = mu + P J^T [ (I - W G)^-1 W (J mu - G q) - q ] (`downdate`) -- and
//     then runs the same formulas at the new point with W = 0, q = 0.  ~480 fp64 instructions per factor instead of ~1090 for
//     the two LDL^T eliminations of rounds 1-3, a third fewer live registers; checked sweep by sweep against the dense reference
//     maths in tests/woodbury_proto.py (numpy) and tests/test_factor_math_host.py (this header compiled for the host).
//   * symmetric matrices are stored packed (upper triangle, row-major): 6x6 -> 21, 3x3 -> 6.
//   * S = Lambda of a belief is SPD: unpivoted LDL^T on packed storage (spd_solve / spd_solve_inverse).
#pragma once
#include <hip/hip_runtime.h>

namespace gbp {

#define GBP_DEV __device__ __forceinline__
#define GBP_HD __host__ __device__ __forceinline__     // the per-factor maths also compiles for the host (tests/host_math.hip: unit tests, no product path)

// 1/x for the pivots and depths of this path (normal, far from over/underflow): hardware seed (v_rcp_f64, ~26 bits)
// + two Newton steps = full double precision (<= 1 ulp) in 5 dependent instructions; the IEEE division sequence
// (v_div_scale / v_div_fmas / v_div_fixup) is twice as long and sits on every elimination's critical path.
GBP_HD double rcp(double x)
{
#if defined(GBP_IEEE_DIV) || !defined(__HIP_DEVICE_COMPILE__)
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
GBP_HD void ldl_factor(double (&a)[Sym<N>::size], double (&invd)[N])
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
GBP_HD void ldl_forward(const double (&a)[Sym<N>::size], double (&b)[N])
{
#pragma unroll
    for (int i = 1; i < N; ++i)
#pragma unroll
        for (int k = 0; k < i; ++k) b[i] -= a[Sym<N>::at(k, i)] * b[k];
}

// b <- L^-T b
template <int N>
GBP_HD void ldl_backward(const double (&a)[Sym<N>::size], double (&b)[N])
{
#pragma unroll
    for (int i = N - 2; i >= 0; --i)
#pragma unroll
        for (int k = i + 1; k < N; ++k) b[i] -= a[Sym<N>::at(i, k)] * b[k];
}

// mu = Lambda^-1 eta for a packed SPD Lambda (VariableNode.update_belief gbp.py:192-193, and the
// belief means of gbp.py:74).  Lambda is consumed.
template <int N>
GBP_HD void spd_solve(double (&lam)[Sym<N>::size], const double (&eta)[N], double (&mu)[N])
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
GBP_HD void spd_inverse(double (&lam)[Sym<N>::size], double (&sig)[Sym<N>::size])
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

// mu = Lambda^-1 eta AND Sigma = Lambda^-1 from one factorisation (a belief in the form the factors read it: mean | covariance)
template <int N>
GBP_HD void spd_solve_inverse(double (&lam)[Sym<N>::size], const double (&eta)[N], double (&mu)[N], double (&sig)[Sym<N>::size])
{
    double invd[N];
    ldl_factor<N>(lam, invd);
#pragma unroll
    for (int i = 0; i < N; ++i) mu[i] = eta[i];
    ldl_forward<N>(lam, mu);
#pragma unroll
    for (int i = 0; i < N; ++i) mu[i] *= invd[i];
    ldl_backward<N>(lam, mu);
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
GBP_HD double sconst(double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(c));
#endif
    return c;
}

GBP_HD void sincos_theta(double x, double &sn, double &cs)
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

GBP_HD Rot rodrigues(double w0, double w1, double w2)
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
GBP_HD void project(const double (&x)[9], const Intrinsics &K, double (&h)[2])
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
GBP_HD void linearise(const double (&x)[9], const Intrinsics &K, double (&Jc)[2][6], double (&Jl)[2][3],
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
GBP_HD double robust_variance(int loss, double sigma2, double nstds, double r0, double r1, bool &flag)
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

template <int N>
GBP_HD void rank2_update(double (&T)[Sym<N>::size], const double (&j0)[N], const double (&j1)[N], const double (&Q)[3], double sign);

// ---- covariance form of Factor.compute_messages (gbp.py:334-373); derivation in the header of this file ----
// The Jacobian block of a variable is given by its two rows j0, j1.  For the camera block two entries are structurally zero
// (d u / d t_y = d v / d t_x = 0, reprojection.py:36-38): skipped at compile time.
template <int N>
constexpr __host__ __device__ bool jzero(int row, int i) { return N == 6 && ((row == 0 && i == 1) || (row == 1 && i == 0)); }

// pj0 = P j0, pj1 = P j1 (P packed symmetric), G = J P J^T as {G00, G01, G11}
template <int N>
GBP_HD void gram(const double (&P)[Sym<N>::size], const double (&j0)[N], const double (&j1)[N], double (&G)[3], double (&pj0)[N], double (&pj1)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double pij = P[i <= j ? Sym<N>::at(i, j) : Sym<N>::at(j, i)];
            if (!jzero<N>(0, j)) a += pij * j0[j];
            if (!jzero<N>(1, j)) b += pij * j1[j];
        }
        pj0[i] = a; pj1[i] = b;
    }
    double g00 = 0.0, g01 = 0.0, g11 = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (!jzero<N>(0, i)) { g00 += j0[i] * pj0[i]; g01 += j0[i] * pj1[i]; }
        if (!jzero<N>(1, i)) g11 += j1[i] * pj1[i];
    }
    G[0] = g00; G[1] = g01; G[2] = g11;
}

// G = J P J^T alone: a row of P J^T is used as soon as it is made (nothing but the three sums stays live)
template <int N>
GBP_HD void gram_only(const double (&P)[Sym<N>::size], const double (&j0)[N], const double (&j1)[N], double (&G)[3])
{
    double g00 = 0.0, g01 = 0.0, g11 = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double pij = P[i <= j ? Sym<N>::at(i, j) : Sym<N>::at(j, i)];
            if (!jzero<N>(0, j)) a += pij * j0[j];
            if (!jzero<N>(1, j)) b += pij * j1[j];
        }
        if (!jzero<N>(0, i)) { g00 += j0[i] * a; g01 += j0[i] * b; }
        if (!jzero<N>(1, i)) g11 += j1[i] * b;
    }
    G[0] = g00; G[1] = g01; G[2] = g11;
}

template <int N>
GBP_HD void jdot(const double (&j0)[N], const double (&j1)[N], const double (&x)[N], double (&m)[2])
{
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (!jzero<N>(0, i)) a += j0[i] * x[i];
        if (!jzero<N>(1, i)) b += j1[i] * x[i];
    }
    m[0] = a; m[1] = b;
}

// The message that eliminating a variable leaves for the other one:
//   (P, mu) covariance and mean of the eliminated variable's belief, (W, q) core and eta coefficients of this factor's OLD
//   message to it (made with the SAME Jacobian rows j0, j1), rho = J x0 + z - h, s = 1 / variance.
//   core = s (I - W G)(I + Q G)^-1 (symmetric),  r = s (I + G Q)^-1 [rho - J mu + G (q - W rho)],  Q = s I - W.
template <int N>
GBP_HD void eliminate(const double (&P)[Sym<N>::size], const double (&mu)[N], const double (&j0)[N], const double (&j1)[N],
                      const double (&W)[3], const double (&q)[2], const double (&rho)[2], double s, double (&core)[3], double (&r)[2])
{
    double G[3], m[2];
    gram_only<N>(P, j0, j1, G);
    jdot<N>(j0, j1, mu, m);
    // E = I - W G,  D = I + Q G = E + s G   (general 2x2)
    const double e00 = 1.0 - (W[0] * G[0] + W[1] * G[1]), e01 = -(W[0] * G[1] + W[1] * G[2]);
    const double e10 = -(W[1] * G[0] + W[2] * G[1]), e11 = 1.0 - (W[1] * G[1] + W[2] * G[2]);
    const double d00 = e00 + s * G[0], d01 = e01 + s * G[1], d10 = e10 + s * G[1], d11 = e11 + s * G[2];
    const double idet = rcp(d00 * d11 - d01 * d10);
    const double i00 = d11 * idet, i01 = -d01 * idet, i10 = -d10 * idet, i11 = d00 * idet;      // D^-1
    core[0] = s * (e00 * i00 + e01 * i10);
    core[1] = 0.5 * s * ((e00 * i01 + e01 * i11) + (e10 * i00 + e11 * i10));
    core[2] = s * (e10 * i01 + e11 * i11);
    const double t0 = q[0] - (W[0] * rho[0] + W[1] * rho[1]), t1 = q[1] - (W[1] * rho[0] + W[2] * rho[1]);
    const double b0 = (rho[0] - m[0]) + (G[0] * t0 + G[1] * t1), b1 = (rho[1] - m[1]) + (G[1] * t0 + G[2] * t1);
    r[0] = s * (i00 * b0 + i10 * b1);                       // (I + G Q)^-1 = (D^-1)^T
    r[1] = s * (i01 * b0 + i11 * b1);
}

// belief minus this factor's old message, in covariance form:  P <- (Lambda - J^T W J)^-1,  mu <- P (eta - J^T q)
template <int N>
GBP_HD void downdate(double (&P)[Sym<N>::size], double (&mu)[N], const double (&j0)[N], const double (&j1)[N], const double (&W)[3], const double (&q)[2])
{
    double G[3], pj0[N], pj1[N], m[2];
    gram<N>(P, j0, j1, G, pj0, pj1);
    jdot<N>(j0, j1, mu, m);
    const double e00 = 1.0 - (W[0] * G[0] + W[1] * G[1]), e01 = -(W[0] * G[1] + W[1] * G[2]);
    const double e10 = -(W[1] * G[0] + W[2] * G[1]), e11 = 1.0 - (W[1] * G[1] + W[2] * G[2]);
    const double idet = rcp(e00 * e11 - e01 * e10);
    const double i00 = e11 * idet, i01 = -e01 * idet, i10 = -e10 * idet, i11 = e00 * idet;      // (I - W G)^-1
    double A[3];                                            // A = (I - W G)^-1 W, symmetric
    A[0] = i00 * W[0] + i01 * W[1];
    A[1] = 0.5 * ((i00 * W[1] + i01 * W[2]) + (i10 * W[0] + i11 * W[1]));
    A[2] = i10 * W[1] + i11 * W[2];
    const double v0 = m[0] - (G[0] * q[0] + G[1] * q[1]), v1 = m[1] - (G[1] * q[0] + G[2] * q[1]);
    const double t0 = A[0] * v0 + A[1] * v1 - q[0], t1 = A[1] * v0 + A[2] * v1 - q[1];
#pragma unroll
    for (int i = 0; i < N; ++i) mu[i] += pj0[i] * t0 + pj1[i] * t1;
    rank2_update<N>(P, pj0, pj1, A, 1.0);
}

// c0 * j0[i] + c1 * j1[i] without the structurally zero term (i is a compile-time constant after unrolling)
template <int N>
GBP_HD double jcomb(int i, double c0, double c1, const double (&j0)[N], const double (&j1)[N])
{
    if (jzero<N>(0, i)) return c1 * j1[i];
    if (jzero<N>(1, i)) return c0 * j0[i];
    return c0 * j0[i] + c1 * j1[i];
}

// dense form of a message for the belief sums: e = J^T q, M = J^T Q J (packed)
template <int N>
GBP_HD void dense_message(const double (&j0)[N], const double (&j1)[N], const double (&q)[2], const double (&Q)[3], double (&e)[N],
                          double (&M)[Sym<N>::size])
{
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double a = jcomb<N>(j, Q[0], Q[1], j0, j1), b = jcomb<N>(j, Q[1], Q[2], j0, j1);
        e[j] = jcomb<N>(j, q[0], q[1], j0, j1);
#pragma unroll
        for (int i = 0; i <= j; ++i) M[Sym<N>::at(i, j)] = jcomb<N>(i, a, b, j0, j1);
    }
}

// A message precision never leaves the span of its Jacobian block: M = J^T Q J with a symmetric 2x2 core Q (W for the
// camera message, V for the landmark message; Lambda is never damped, gbp.py:368).  Only Q (3 doubles) is stored per
// message instead of the packed 21 / 6; the dense matrix is rebuilt from the Jacobian at the linearisation point the
// message was computed with.   T += sign * J^T Q J   on packed upper storage, J given by its two rows.
template <int N>
GBP_HD void rank2_update(double (&T)[Sym<N>::size], const double (&j0)[N], const double (&j1)[N], const double (&Q)[3],
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
GBP_HD double factor_lambda_max(const double (&Jc)[2][6], const double (&Jl)[2][3], double s)
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
