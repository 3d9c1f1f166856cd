// host_math.hip -- TEST INFRASTRUCTURE: the per-factor maths of the sweep kernels (gbp_amd/csrc/gbp_math.hpp, factor_core of
// gbp_kernels.hpp) compiled for the HOST, so that the exact C++ the GPU runs can be driven factor by factor from a CPU test and
// compared with the reference-style dense maths before it ever reaches a GPU.  Built by tests/test_factor_math_host.py with
// hipcc (the device pass rides along unused); nothing in the product links or loads it (the product has no CPU path).
#include "../../gbp_amd/csrc/gbp_kernels.hpp"

using namespace gbp;

template <int LOSS, bool XTRA>
static void run(int n, const Params &p, double *x0, const double *z, int *st, double *avar, const double *muC, const double *PC, const double *muL,
                const double *PL, double *qC, double *qL, double *WC, double *VL, double *eC, double *MC, double *eL, double *ML, double *xt, int *relin)
{
    for (int f = 0; f < n; ++f) {
        double x[9], zz[2] = {z[2 * f], z[2 * f + 1]}, mc[6], pc[21], ml[3], qc[2], ql[2], w[3], v[3], ec[6], el[3], mcn[21], mln[6];
        for (int k = 0; k < 9; ++k) x[k] = x0[9 * f + k];
        for (int k = 0; k < 6; ++k) mc[k] = muC[6 * f + k];
        for (int k = 0; k < 21; ++k) pc[k] = PC[21 * f + k];
        for (int k = 0; k < 3; ++k) ml[k] = muL[3 * f + k];
        for (int k = 0; k < 2; ++k) { qc[k] = qC[2 * f + k]; ql[k] = qL[2 * f + k]; }
        for (int k = 0; k < 3; ++k) { w[k] = WC[3 * f + k]; v[k] = VL[3 * f + k]; }
        const double *pl = PL + 6 * f;
        double *xo = x0 + 9 * f;
        relin[f] = factor_core<LOSS, XTRA>(p, x, zz, st[f], avar[f], mc, pc, ml,
                                           [pl](double (&c)[6]) { for (int k = 0; k < 6; ++k) c[k] = pl[k]; },
                                           [xo](const double (&xn)[9]) { for (int k = 0; k < 9; ++k) xo[k] = xn[k]; },
                                           qc, ql, w, v, ec, el, mcn, mln, XTRA ? xt + 9 * f : nullptr) ? 1 : 0;
        for (int k = 0; k < 2; ++k) { qC[2 * f + k] = qc[k]; qL[2 * f + k] = ql[k]; }
        for (int k = 0; k < 3; ++k) { WC[3 * f + k] = w[k]; VL[3 * f + k] = v[k]; }
        for (int k = 0; k < 6; ++k) eC[6 * f + k] = ec[k];
        for (int k = 0; k < 21; ++k) MC[21 * f + k] = mcn[k];
        for (int k = 0; k < 3; ++k) eL[3 * f + k] = el[k];
        for (int k = 0; k < 6; ++k) ML[6 * f + k] = mln[k];
    }
}

extern "C" {

// one sweep's per-factor part (robustify, relinearisation test, linearisation, both messages) over n factors; st = the state words
int hm_sweep_factors(int n, const double *K4, double sigma2, double nstds, double beta, double eta_damping, int num_undamped, int min_linear,
                     int loss, int robustify, int local_relin, int stage, int clk, int clk_inc, double *x0, const double *z, int *st, double *avar, const double *muC,
                     const double *PC, const double *muL, const double *PL, double *qC, double *qL, double *WC, double *VL, double *eC, double *MC,
                     double *eL, double *ML, double *xt, int *relin)
{
    Params p{};
    p.K = Intrinsics{K4[0], K4[1], K4[2], K4[3]};
    p.sigma2 = sigma2; p.nstds = nstds; p.beta = beta; p.eta_damping = eta_damping;
    p.num_undamped = num_undamped; p.min_linear = min_linear; p.loss = loss; p.robustify = robustify; p.local_relin = local_relin; p.stage = stage;
    p.clk = clk; p.clk_inc = clk_inc;                        // the relinearisation clock after this call, and whether the call advances it
#define GO(L, X) run<L, X>(n, p, x0, z, st, avar, muC, PC, muL, PL, qC, qL, WC, VL, eC, MC, eL, ML, xt, relin)
    if (xt) { if (loss == 0) GO(0, true); else if (loss == 1) GO(1, true); else GO(2, true); }
    else { if (loss == 0) GO(0, false); else if (loss == 1) GO(1, false); else GO(2, false); }
#undef GO
    return 0;
}

// beliefs in the form the factors read them: mu = Lambda^-1 eta, Sigma = Lambda^-1 (packed), dofs = 3 or 6
int hm_belief(int dofs, int n, const double *eta, const double *lam, double *mu, double *sig)
{
    for (int v = 0; v < n; ++v) {
        if (dofs == 6) {
            double l[21], e[6], m[6], s[21];
            for (int k = 0; k < 21; ++k) l[k] = lam[21 * v + k];
            for (int k = 0; k < 6; ++k) e[k] = eta[6 * v + k];
            spd_solve_inverse<6>(l, e, m, s);
            for (int k = 0; k < 6; ++k) mu[6 * v + k] = m[k];
            for (int k = 0; k < 21; ++k) sig[21 * v + k] = s[k];
        } else {
            double l[6], e[3], m[3], s[6];
            for (int k = 0; k < 6; ++k) l[k] = lam[6 * v + k];
            for (int k = 0; k < 3; ++k) e[k] = eta[3 * v + k];
            spd_solve_inverse<3>(l, e, m, s);
            for (int k = 0; k < 3; ++k) mu[3 * v + k] = m[k];
            for (int k = 0; k < 6; ++k) sig[6 * v + k] = s[k];
        }
    }
    return 0;
}

int hm_state_pack(int iters, int clk, int rank, int robust, int damped, int pending) { return state_pack(iters, clk, rank, robust != 0, damped != 0, pending != 0); }
int hm_state_age(int st, int clk) { return state_age(st, clk); }
int hm_state_set_age(int st, int iters, int clk) { return state_set_age(st, iters, clk); }

}  // extern "C"
