/*
 * gbp_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, CPU, fp64 restatement of the reference's GBP bundle-adjustment hot path
 * (joeaortiz/gbp: gbp/gbp.py, gbp/gbp_ba.py, gbp/factors/reprojection.py, utils/derivatives.py,
 * utils/lie_algebra.py).  It exists to CHECK the HIP engine (tests/, __graft_entry__.smoke())
 * and to be TIMED beside it (bench.py cpu_baseline, kind "port").  Nothing under gbp_amd/ may
 * import, link or call it.
 *
 * Parity pin: this file is checked against golden vectors produced by importing the reference
 * itself (tests/golden/make_golden.py -> G1..G9, tests/test_oracle_golden.py).
 *
 * It deliberately keeps the reference's data model: dense 9x9 factor precision stored per factor
 * and rescaled in place by robustify, dense unsymmetrised 6x6 / 3x3 messages, general (LU,
 * partial pivoting) inverses like numpy.linalg.inv, reference factor order (camera-major) and
 * the reference's left-to-right matrix product order.  Every function cites what it follows.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { LOSS_NONE = 0, LOSS_HUBER = 1, LOSS_CONSTANT = 2 };

typedef struct gbpo {
    int C, L, F;
    double K[9];
    double sigma2;       /* Factor.gauss_noise_var            gbp/gbp.py:236 */
    int loss;            /* Factor.loss                       gbp/gbp.py:243 */
    double nstds;        /* Factor.mahalanobis_threshold      gbp/gbp.py:244 */
    double beta;         /* FactorGraph.beta                  gbp/gbp.py:32  */
    int num_undamped;    /* FactorGraph.num_undamped_iters    gbp/gbp.py:33  */
    int min_linear;      /* FactorGraph.min_linear_iters      gbp/gbp.py:34  */
    double eta_damping;  /* FactorGraph.eta_damping           gbp/gbp.py:28  */
    int nthreads;
    /* variables: cameras then landmarks (gbp/gbp_ba.py:114-125) */
    double *cam_mu, *lmk_mu;
    double *cam_prior_eta, *cam_prior_lam, *lmk_prior_eta, *lmk_prior_lam;
    double *cam_bel_eta, *cam_bel_lam, *lmk_bel_eta, *lmk_bel_lam;
    /* factors in reference order (gbp/gbp_ba.py:128-130) */
    int *f_cam, *f_lmk;
    double *z, *f_eta, *f_lam, *linpoint;
    double *m_cam_eta, *m_cam_lam, *m_lmk_eta, *m_lmk_lam;
    double *adaptive_var, *damping;
    int *iters;
    unsigned char *robust;
    /* VariableNode.adj_factors as CSR (factor ids ascending == append order) */
    int *cam_ptr, *cam_adj, *lmk_ptr, *lmk_adj;
    double *scratch;     /* F doubles for ordered reductions */
} gbpo_t;

/* ---------------------------------------------------------------- small dense helpers -- */

/* numpy.linalg.inv restated: Gauss-Jordan with partial pivoting on [A | I]. */
static void inv_n(const double *A, int n, double *out)
{
    double a[36], b[36];
    for (int i = 0; i < n * n; ++i) { a[i] = A[i]; b[i] = 0.0; }
    for (int i = 0; i < n; ++i) b[i * n + i] = 1.0;
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = fabs(a[c * n + c]);
        for (int r = c + 1; r < n; ++r)
            if (fabs(a[r * n + c]) > best) { best = fabs(a[r * n + c]); p = r; }
        if (p != c)
            for (int k = 0; k < n; ++k) {
                double t = a[c * n + k]; a[c * n + k] = a[p * n + k]; a[p * n + k] = t;
                t = b[c * n + k]; b[c * n + k] = b[p * n + k]; b[p * n + k] = t;
            }
        double d = 1.0 / a[c * n + c];
        for (int k = 0; k < n; ++k) { a[c * n + k] *= d; b[c * n + k] *= d; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = a[r * n + c];
            if (f == 0.0) continue;
            for (int k = 0; k < n; ++k) { a[r * n + k] -= f * a[c * n + k]; b[r * n + k] -= f * b[c * n + k]; }
        }
    }
    for (int i = 0; i < n * n; ++i) out[i] = b[i];
}

/* C(m x n) = A(m x k) @ B(k x n), row-major */
static void matmul(const double *A, const double *B, double *Cm, int m, int k, int n)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int t = 0; t < k; ++t) s += A[i * k + t] * B[t * n + j];
            Cm[i * n + j] = s;
        }
}

/* utils/lie_algebra.py:11-17 */
static void hat3(const double *x, double *H)
{
    H[0] = 0.0;   H[1] = -x[2]; H[2] = x[1];
    H[3] = x[2];  H[4] = 0.0;   H[5] = -x[0];
    H[6] = -x[1]; H[7] = x[0];  H[8] = 0.0;
}

/* utils/lie_algebra.py:32-42 */
static void so3exp(const double *w, double *R)
{
    double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int i = 0; i < 9; ++i) R[i] = 0.0;
    R[0] = R[4] = R[8] = 1.0;
    if (theta < DBL_EPSILON * 3) return;
    double H[9], H2[9];
    hat3(w, H);
    matmul(H, H, H2, 3, 3, 3);
    double a = sin(theta) / theta, b = (1.0 - cos(theta)) / (theta * theta);
    for (int i = 0; i < 9; ++i) R[i] = R[i] + a * H[i] + b * H2[i];
}

/* gbp/factors/reprojection.py:12-24 with utils/transformations.py:5-7 */
static void meas_fn(const double *x, const double *K, double *h)
{
    double R[9], p[3], q[3];
    so3exp(x + 3, R);
    for (int i = 0; i < 3; ++i)
        p[i] = (R[i * 3] * x[6] + R[i * 3 + 1] * x[7] + R[i * 3 + 2] * x[8]) + x[i];
    for (int i = 0; i < 3; ++i)
        q[i] = K[i * 3] * p[0] + K[i * 3 + 1] * p[1] + K[i * 3 + 2] * p[2];
    h[0] = q[0] / q[2];
    h[1] = q[1] / q[2];
}

/* utils/derivatives.py:36-45 */
static void dR_wx_dw(const double *w, const double *x, double *D)
{
    double R[9], Xh[9], Wh[9], RX[9], Rt_I[9], M[9], T[9];
    so3exp(w, R);
    hat3(x, Xh);
    hat3(w, Wh);
    matmul(R, Xh, RX, 3, 3, 3);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rt_I[i * 3 + j] = R[j * 3 + i] - (i == j ? 1.0 : 0.0);
    matmul(Rt_I, Wh, M, 3, 3, 3);
    double ww = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i * 3 + j] = (w[i] * w[j] + M[i * 3 + j]) / ww;
    matmul(RX, M, T, 3, 3, 3);
    for (int i = 0; i < 9; ++i) D[i] = -T[i];
}

/* gbp/factors/reprojection.py:27-44 with utils/derivatives.py:48-50 */
static void jac_fn(const double *x, const double *K, double *J /* 2x9 */)
{
    double R[9], p[3], q[3], Jp[6], JK[6], D[9], A[6], B[6];
    so3exp(x + 3, R);
    for (int i = 0; i < 3; ++i)
        p[i] = (R[i * 3] * x[6] + R[i * 3 + 1] * x[7] + R[i * 3 + 2] * x[8]) + x[i];
    for (int i = 0; i < 3; ++i)
        q[i] = K[i * 3] * p[0] + K[i * 3 + 1] * p[1] + K[i * 3 + 2] * p[2];
    Jp[0] = 1.0 / q[2]; Jp[1] = 0.0 / q[2]; Jp[2] = -q[0] / (q[2] * q[2]);
    Jp[3] = 0.0 / q[2]; Jp[4] = 1.0 / q[2]; Jp[5] = -q[1] / (q[2] * q[2]);
    matmul(Jp, K, JK, 2, 3, 3);
    dR_wx_dw(x + 3, x + 6, D);
    matmul(JK, D, A, 2, 3, 3);
    matmul(JK, R, B, 2, 3, 3);
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) {
            J[r * 9 + c] = JK[r * 3 + c];
            J[r * 9 + 3 + c] = A[r * 3 + c];
            J[r * 9 + 6 + c] = B[r * 3 + c];
        }
}

/* Factor.compute_factor, array-measurement branch: gbp/gbp.py:278-294 */
static void compute_factor(gbpo_t *g, int f, const double *linpoint)
{
    double *lp = g->linpoint + 9 * (size_t)f;
    double x[9];
    for (int i = 0; i < 9; ++i) x[i] = linpoint[i];
    for (int i = 0; i < 9; ++i) lp[i] = x[i];
    double J[18], h[2], JtM[18], r[2];
    jac_fn(x, g->K, J);
    meas_fn(x, g->K, h);
    double ml = 1.0 / g->adaptive_var[f];            /* eye(2) / adaptive var */
    for (int i = 0; i < 9; ++i)
        for (int k = 0; k < 2; ++k) JtM[i * 2 + k] = J[k * 9 + i] * ml;   /* J.T @ meas_model_lambda */
    double *lam = g->f_lam + 81 * (size_t)f, *eta = g->f_eta + 9 * (size_t)f;
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j)
            lam[i * 9 + j] = JtM[i * 2] * J[j] + JtM[i * 2 + 1] * J[9 + j];
    for (int k = 0; k < 2; ++k) {
        double s = 0.0;
        for (int i = 0; i < 9; ++i) s += J[k * 9 + i] * x[i];
        r[k] = s + g->z[2 * (size_t)f + k] - h[k];
    }
    for (int i = 0; i < 9; ++i) eta[i] = JtM[i * 2] * r[0] + JtM[i * 2 + 1] * r[1];
}

/* mean of one adjacent belief: inv(lam) @ eta   (gbp/gbp.py:74, :257, :308) */
static void belief_mean(const double *eta, const double *lam, int n, double *mu)
{
    double S[36];
    inv_n(lam, n, S);
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += S[i * n + j] * eta[j];
        mu[i] = s;
    }
}

static void adj_means(const gbpo_t *g, int f, double *m9)
{
    int c = g->f_cam[f], l = g->f_lmk[f];
    belief_mean(g->cam_bel_eta + 6 * (size_t)c, g->cam_bel_lam + 36 * (size_t)c, 6, m9);
    belief_mean(g->lmk_bel_eta + 3 * (size_t)l, g->lmk_bel_lam + 9 * (size_t)l, 3, m9 + 6);
}

/* ------------------------------------------------------------------------- life cycle -- */

#define ALLOC(p, n) do { (p) = calloc((size_t)(n) > 0 ? (size_t)(n) : 1, sizeof(*(p))); if (!(p)) return NULL; } while (0)

/* create_ba_graph: gbp/gbp_ba.py:97-150.  Inputs are in FILE order; factors are created
 * camera-major by the O(C*F) scan at :128-130 == stable sort by camera id. */
gbpo_t *gbpo_create(int C, int L, int F, const double *K4, const double *cam_means,
                    const double *lmk_means, const double *meas, const int *cam_idx,
                    const int *lmk_idx, double sigma, int loss, double nstds, double beta,
                    int num_undamped, int min_linear, double eta_damping)
{
    gbpo_t *g = calloc(1, sizeof(gbpo_t));
    if (!g) return NULL;
    g->C = C; g->L = L; g->F = F;
    g->K[0] = K4[0]; g->K[4] = K4[1]; g->K[2] = K4[2]; g->K[5] = K4[3]; g->K[8] = 1.0;  /* read_balfile.py:14-16 */
    g->sigma2 = sigma * sigma;
    g->loss = loss; g->nstds = nstds; g->beta = beta;
    g->num_undamped = num_undamped; g->min_linear = min_linear; g->eta_damping = eta_damping;
    g->nthreads = 1;
    ALLOC(g->cam_mu, 6 * (size_t)C); ALLOC(g->lmk_mu, 3 * (size_t)L);
    ALLOC(g->cam_prior_eta, 6 * (size_t)C); ALLOC(g->cam_prior_lam, 36 * (size_t)C);
    ALLOC(g->lmk_prior_eta, 3 * (size_t)L); ALLOC(g->lmk_prior_lam, 9 * (size_t)L);
    ALLOC(g->cam_bel_eta, 6 * (size_t)C); ALLOC(g->cam_bel_lam, 36 * (size_t)C);
    ALLOC(g->lmk_bel_eta, 3 * (size_t)L); ALLOC(g->lmk_bel_lam, 9 * (size_t)L);
    ALLOC(g->f_cam, F); ALLOC(g->f_lmk, F); ALLOC(g->z, 2 * (size_t)F);
    ALLOC(g->f_eta, 9 * (size_t)F); ALLOC(g->f_lam, 81 * (size_t)F); ALLOC(g->linpoint, 9 * (size_t)F);
    ALLOC(g->m_cam_eta, 6 * (size_t)F); ALLOC(g->m_cam_lam, 36 * (size_t)F);
    ALLOC(g->m_lmk_eta, 3 * (size_t)F); ALLOC(g->m_lmk_lam, 9 * (size_t)F);
    ALLOC(g->adaptive_var, F); ALLOC(g->damping, F); ALLOC(g->iters, F); ALLOC(g->robust, F);
    ALLOC(g->cam_ptr, C + 1); ALLOC(g->cam_adj, F); ALLOC(g->lmk_ptr, L + 1); ALLOC(g->lmk_adj, F);
    ALLOC(g->scratch, F);
    memcpy(g->cam_mu, cam_means, sizeof(double) * 6 * (size_t)C);
    memcpy(g->lmk_mu, lmk_means, sizeof(double) * 3 * (size_t)L);

    /* camera-major stable counting sort of the observations */
    for (int i = 0; i < F; ++i) {
        if (cam_idx[i] < 0 || cam_idx[i] >= C || lmk_idx[i] < 0 || lmk_idx[i] >= L) { return NULL; }
        g->cam_ptr[cam_idx[i] + 1]++;
    }
    for (int c = 0; c < C; ++c) g->cam_ptr[c + 1] += g->cam_ptr[c];
    {
        int *cur = malloc(sizeof(int) * (size_t)(C > 0 ? C : 1));
        for (int c = 0; c < C; ++c) cur[c] = g->cam_ptr[c];
        for (int i = 0; i < F; ++i) {
            int f = cur[cam_idx[i]]++;
            g->f_cam[f] = cam_idx[i]; g->f_lmk[f] = lmk_idx[i];
            g->z[2 * (size_t)f] = meas[2 * (size_t)i]; g->z[2 * (size_t)f + 1] = meas[2 * (size_t)i + 1];
        }
        free(cur);
    }
    for (int f = 0; f < F; ++f) g->cam_adj[f] = f;                       /* cam c owns [cam_ptr[c], cam_ptr[c+1]) */
    for (int f = 0; f < F; ++f) g->lmk_ptr[g->f_lmk[f] + 1]++;
    for (int l = 0; l < L; ++l) g->lmk_ptr[l + 1] += g->lmk_ptr[l];
    {
        int *cur = malloc(sizeof(int) * (size_t)(L > 0 ? L : 1));
        for (int l = 0; l < L; ++l) cur[l] = g->lmk_ptr[l];
        for (int f = 0; f < F; ++f) g->lmk_adj[cur[g->f_lmk[f]]++] = f; /* ascending factor id == append order :138-139 */
        free(cur);
    }
    /* Factor.__init__ defaults gbp/gbp.py:236-249, then compute_factor at the file means gbp_ba.py:136-137 */
    for (int f = 0; f < F; ++f) {
        g->adaptive_var[f] = g->sigma2;
        g->damping[f] = 0.0;
        g->iters[f] = 1;
        g->robust[f] = 0;
        double lp[9];
        for (int i = 0; i < 6; ++i) lp[i] = g->cam_mu[6 * (size_t)g->f_cam[f] + i];
        for (int i = 0; i < 3; ++i) lp[6 + i] = g->lmk_mu[3 * (size_t)g->f_lmk[f] + i];
        compute_factor(g, f, lp);
    }
    return g;
}

void gbpo_destroy(gbpo_t *g)
{
    if (!g) return;
    free(g->cam_mu); free(g->lmk_mu);
    free(g->cam_prior_eta); free(g->cam_prior_lam); free(g->lmk_prior_eta); free(g->lmk_prior_lam);
    free(g->cam_bel_eta); free(g->cam_bel_lam); free(g->lmk_bel_eta); free(g->lmk_bel_lam);
    free(g->f_cam); free(g->f_lmk); free(g->z); free(g->f_eta); free(g->f_lam); free(g->linpoint);
    free(g->m_cam_eta); free(g->m_cam_lam); free(g->m_lmk_eta); free(g->m_lmk_lam);
    free(g->adaptive_var); free(g->damping); free(g->iters); free(g->robust);
    free(g->cam_ptr); free(g->cam_adj); free(g->lmk_ptr); free(g->lmk_adj); free(g->scratch);
    free(g);
}

void gbpo_set_threads(gbpo_t *g, int n) { g->nthreads = n > 0 ? n : 1; }

/* ---------------------------------------------------------------------------- priors -- */

/* BAFactorGraph.generate_priors_var: gbp/gbp_ba.py:20-34 */
void gbpo_generate_priors(gbpo_t *g, double weaker_factor)
{
    for (int v = 0; v < g->C + g->L; ++v) {
        int is_cam = v < g->C, i = is_cam ? v : v - g->C, n = is_cam ? 6 : 3;
        const int *ptr = is_cam ? g->cam_ptr : g->lmk_ptr, *adj = is_cam ? g->cam_adj : g->lmk_adj;
        double mx = 0.0;
        for (int e = ptr[i]; e < ptr[i + 1]; ++e) {
            const double *lam = g->f_lam + 81 * (size_t)adj[e];
            double fm = lam[0];
            for (int k = 1; k < 81; ++k) if (lam[k] > fm) fm = lam[k];      /* np.max over the whole 9x9 */
            if (fm > mx) mx = fm;
        }
        double lp = mx / (weaker_factor * weaker_factor);
        double *pl = is_cam ? g->cam_prior_lam + 36 * (size_t)i : g->lmk_prior_lam + 9 * (size_t)i;
        double *pe = is_cam ? g->cam_prior_eta + 6 * (size_t)i : g->lmk_prior_eta + 3 * (size_t)i;
        const double *mu = is_cam ? g->cam_mu + 6 * (size_t)i : g->lmk_mu + 3 * (size_t)i;
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) pl[a * n + b] = (a == b ? 1.0 : 0.0) * mx / (weaker_factor * weaker_factor);
        (void)lp;
        for (int a = 0; a < n; ++a) {
            double s = 0.0;
            for (int b = 0; b < n; ++b) s += pl[a * n + b] * mu[b];
            pe[a] = s;
        }
    }
}

/* BAFactorGraph.weaken_priors: gbp/gbp_ba.py:36-42 */
void gbpo_weaken_priors(gbpo_t *g, double factor)
{
    for (size_t i = 0; i < 6 * (size_t)g->C; ++i) g->cam_prior_eta[i] *= factor;
    for (size_t i = 0; i < 36 * (size_t)g->C; ++i) g->cam_prior_lam[i] *= factor;
    for (size_t i = 0; i < 3 * (size_t)g->L; ++i) g->lmk_prior_eta[i] *= factor;
    for (size_t i = 0; i < 9 * (size_t)g->L; ++i) g->lmk_prior_lam[i] *= factor;
}

/* BAFactorGraph.set_priors_var: gbp/gbp_ba.py:44-52 (covariances in, cams then landmarks) */
void gbpo_set_priors(gbpo_t *g, const double *cam_cov, const double *lmk_cov)
{
    for (int c = 0; c < g->C; ++c) {
        inv_n(cam_cov + 36 * (size_t)c, 6, g->cam_prior_lam + 36 * (size_t)c);
        for (int a = 0; a < 6; ++a) {
            double s = 0.0;
            for (int b = 0; b < 6; ++b) s += g->cam_prior_lam[36 * (size_t)c + a * 6 + b] * g->cam_mu[6 * (size_t)c + b];
            g->cam_prior_eta[6 * (size_t)c + a] = s;
        }
    }
    for (int l = 0; l < g->L; ++l) {
        inv_n(lmk_cov + 9 * (size_t)l, 3, g->lmk_prior_lam + 9 * (size_t)l);
        for (int a = 0; a < 3; ++a) {
            double s = 0.0;
            for (int b = 0; b < 3; ++b) s += g->lmk_prior_lam[9 * (size_t)l + a * 3 + b] * g->lmk_mu[3 * (size_t)l + b];
            g->lmk_prior_eta[3 * (size_t)l + a] = s;
        }
    }
}

/* --------------------------------------------------------------------------- the sweep -- */

/* VariableNode.update_belief: gbp/gbp.py:176-198 (the push at :196-198 is implicit: factors
 * read the variable arrays, which is what the aliasing in the reference amounts to). */
static void update_one(gbpo_t *g, int is_cam, int i)
{
    int n = is_cam ? 6 : 3;
    const int *ptr = is_cam ? g->cam_ptr : g->lmk_ptr, *adj = is_cam ? g->cam_adj : g->lmk_adj;
    double eta[6], lam[36], S[36];
    const double *pe = is_cam ? g->cam_prior_eta + 6 * (size_t)i : g->lmk_prior_eta + 3 * (size_t)i;
    const double *pl = is_cam ? g->cam_prior_lam + 36 * (size_t)i : g->lmk_prior_lam + 9 * (size_t)i;
    for (int a = 0; a < n; ++a) eta[a] = pe[a];
    for (int a = 0; a < n * n; ++a) lam[a] = pl[a];
    for (int e = ptr[i]; e < ptr[i + 1]; ++e) {
        size_t f = (size_t)adj[e];
        const double *me = is_cam ? g->m_cam_eta + 6 * f : g->m_lmk_eta + 3 * f;
        const double *ml = is_cam ? g->m_cam_lam + 36 * f : g->m_lmk_lam + 9 * f;
        for (int a = 0; a < n; ++a) eta[a] += me[a];
        for (int a = 0; a < n * n; ++a) lam[a] += ml[a];
    }
    double *be = is_cam ? g->cam_bel_eta + 6 * (size_t)i : g->lmk_bel_eta + 3 * (size_t)i;
    double *bl = is_cam ? g->cam_bel_lam + 36 * (size_t)i : g->lmk_bel_lam + 9 * (size_t)i;
    double *mu = is_cam ? g->cam_mu + 6 * (size_t)i : g->lmk_mu + 3 * (size_t)i;
    for (int a = 0; a < n; ++a) be[a] = eta[a];
    for (int a = 0; a < n * n; ++a) bl[a] = lam[a];
    inv_n(lam, n, S);
    for (int a = 0; a < n; ++a) {
        double s = 0.0;
        for (int b = 0; b < n; ++b) s += S[a * n + b] * eta[b];
        mu[a] = s;
    }
}

/* FactorGraph.update_all_beliefs: gbp/gbp.py:56-58 */
void gbpo_update_beliefs(gbpo_t *g)
{
    int nv = g->C + g->L;
#pragma omp parallel for schedule(static) num_threads(g->nthreads)
    for (int v = 0; v < nv; ++v) {
        if (v < g->C) update_one(g, 1, v); else update_one(g, 0, v - g->C);
    }
}

/* Factor.robustify_loss: gbp/gbp.py:296-332 (residual at the stored linearisation point :309) */
static void robustify_one(gbpo_t *g, int f)
{
    double old = g->adaptive_var[f];
    if (g->loss == LOSS_NONE) {
        g->adaptive_var[f] = g->sigma2;
    } else {
        double h[2];
        meas_fn(g->linpoint + 9 * (size_t)f, g->K, h);
        double d0 = g->z[2 * (size_t)f] - h[0], d1 = g->z[2 * (size_t)f + 1] - h[1];
        double m = sqrt(d0 * d0 + d1 * d1) / sqrt(g->sigma2);
        if (m > g->nstds) {
            if (g->loss == LOSS_HUBER)
                g->adaptive_var[f] = g->sigma2 * (m * m) / (2 * (g->nstds * m - 0.5 * (g->nstds * g->nstds)));
            else
                g->adaptive_var[f] = m * m;                              /* sic: gbp/gbp.py:324 */
            g->robust[f] = 1;
        } else {
            g->robust[f] = 0;
            g->adaptive_var[f] = g->sigma2;
        }
    }
    double ratio = old / g->adaptive_var[f];
    double *eta = g->f_eta + 9 * (size_t)f, *lam = g->f_lam + 81 * (size_t)f;
    for (int i = 0; i < 9; ++i) eta[i] *= ratio;
    for (int i = 0; i < 81; ++i) lam[i] *= ratio;
}

/* FactorGraph.relinearise_factors: gbp/gbp.py:64-80 */
static void relinearise_one(gbpo_t *g, int f)
{
    double m[9], d = 0.0;
    adj_means(g, f, m);
    const double *lp = g->linpoint + 9 * (size_t)f;
    for (int i = 0; i < 9; ++i) d += (lp[i] - m[i]) * (lp[i] - m[i]);
    if (sqrt(d) > g->beta && g->iters[f] >= g->min_linear) {
        compute_factor(g, f, m);
        g->iters[f] = 0;
        g->damping[f] = 0.0;
    } else {
        g->iters[f] += 1;
    }
}

/* Factor.compute_messages specialised to adj = [camera(6), landmark(3)]: gbp/gbp.py:334-373 */
static void messages_one(gbpo_t *g, int f, double damping)
{
    size_t F = (size_t)f;
    int c = g->f_cam[f], l = g->f_lmk[f];
    const double *fe = g->f_eta + 9 * F, *fl = g->f_lam + 81 * F;
    const double *bce = g->cam_bel_eta + 6 * (size_t)c, *bcl = g->cam_bel_lam + 36 * (size_t)c;
    const double *ble = g->lmk_bel_eta + 3 * (size_t)l, *bll = g->lmk_bel_lam + 9 * (size_t)l;
    double *mce = g->m_cam_eta + 6 * F, *mcl = g->m_cam_lam + 36 * F;
    double *mle = g->m_lmk_eta + 3 * F, *mll = g->m_lmk_lam + 9 * F;
    double new_ce[6], new_cl[36], new_le[3], new_ll[9];

    {   /* v = 0: message to the camera; fold in the landmark's belief minus its old message */
        double eta[9], lam[81];
        memcpy(eta, fe, sizeof eta); memcpy(lam, fl, sizeof lam);
        for (int i = 0; i < 3; ++i) eta[6 + i] += ble[i] - mle[i];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) lam[(6 + i) * 9 + 6 + j] += bll[i * 3 + j] - mll[i * 3 + j];
        double lono[18], lnoo[18], lnono[9], inv[9], t[18], u[36], w[6];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) lono[i * 3 + j] = lam[i * 9 + 6 + j];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) lnoo[i * 6 + j] = lam[(6 + i) * 9 + j];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) lnono[i * 3 + j] = lam[(6 + i) * 9 + 6 + j];
        inv_n(lnono, 3, inv);
        matmul(lono, inv, t, 6, 3, 3);              /* (lono @ inv(lnono)) ... */
        matmul(t, lnoo, u, 6, 3, 6);                /* ... @ lnoo */
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) new_cl[i * 6 + j] = lam[i * 9 + j] - u[i * 6 + j];
        matmul(t, eta + 6, w, 6, 3, 1);
        for (int i = 0; i < 6; ++i) new_ce[i] = (1 - damping) * (eta[i] - w[i]) + damping * mce[i];
    }
    {   /* v = 1: message to the landmark; fold in the camera's belief minus its old message */
        double eta[9], lam[81];
        memcpy(eta, fe, sizeof eta); memcpy(lam, fl, sizeof lam);
        for (int i = 0; i < 6; ++i) eta[i] += bce[i] - mce[i];
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) lam[i * 9 + j] += bcl[i * 6 + j] - mcl[i * 6 + j];
        double lono[18], lnoo[18], lnono[36], inv[36], t[18], u[9], w[3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) lono[i * 6 + j] = lam[(6 + i) * 9 + j];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) lnoo[i * 3 + j] = lam[i * 9 + 6 + j];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) lnono[i * 6 + j] = lam[i * 9 + j];
        inv_n(lnono, 6, inv);
        matmul(lono, inv, t, 3, 6, 6);
        matmul(t, lnoo, u, 3, 6, 3);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) new_ll[i * 3 + j] = lam[(6 + i) * 9 + 6 + j] - u[i * 3 + j];
        matmul(t, eta, w, 3, 6, 1);
        for (int i = 0; i < 3; ++i) new_le[i] = (1 - damping) * (eta[6 + i] - w[i]) + damping * mle[i];
    }
    memcpy(mce, new_ce, sizeof new_ce); memcpy(mcl, new_cl, sizeof new_cl);      /* commit :371-373 */
    memcpy(mle, new_le, sizeof new_le); memcpy(mll, new_ll, sizeof new_ll);
}

void gbpo_robustify(gbpo_t *g)                                   /* gbp/gbp.py:82-84 */
{
#pragma omp parallel for schedule(static) num_threads(g->nthreads)
    for (int f = 0; f < g->F; ++f) robustify_one(g, f);
}

void gbpo_relinearise(gbpo_t *g)                                 /* gbp/gbp.py:64-80 */
{
#pragma omp parallel for schedule(static) num_threads(g->nthreads)
    for (int f = 0; f < g->F; ++f) relinearise_one(g, f);
}

void gbpo_compute_factors(gbpo_t *g)                             /* gbp/gbp.py:60-62: Factor.compute_factor() with no linpoint = the
                                                                    adjacent belief means (gbp.py:273-277), every factor; nothing else changes */
{
#pragma omp parallel for schedule(static) num_threads(g->nthreads)
    for (int f = 0; f < g->F; ++f) {
        double m[9];
        adj_means(g, f, m);
        compute_factor(g, f, m);
    }
}

void gbpo_compute_messages(gbpo_t *g, int local_relin)           /* gbp/gbp.py:46-54 */
{
#pragma omp parallel for schedule(static) num_threads(g->nthreads)
    for (int f = 0; f < g->F; ++f) {
        if (local_relin) {
            if (g->iters[f] == g->num_undamped) g->damping[f] = g->eta_damping;
            messages_one(g, f, g->damping[f]);
        } else {
            messages_one(g, f, g->eta_damping);
        }
    }
}

/* FactorGraph.synchronous_iteration: gbp/gbp.py:86-92 (BA graphs are nonlinear_factors=True) */
void gbpo_iterate(gbpo_t *g, int n_iters, int robustify, int local_relin)
{
    for (int it = 0; it < n_iters; ++it) {
        if (robustify) gbpo_robustify(g);
        if (local_relin) gbpo_relinearise(g);
        gbpo_compute_messages(g, local_relin);
        gbpo_update_beliefs(g);
    }
}

static void residual_one(const gbpo_t *g, int f, double *r);

/* ---------------------------------------------------------- landmark-sharded sweep pieces -- */
/* Test doubles of the C ABI's gbp_ba_shard_begin / gbp_ba_shard_end (include/gbp_ba.h): the reference has no
 * multi-device code; these split update_all_beliefs (gbp/gbp.py:56-58) into "landmarks + this shard's camera
 * message sums" and "cameras from the gathered sums", so the host-side sharding logic can be tested on CPU. */
void gbpo_update_lmk_beliefs(gbpo_t *g)
{
#pragma omp parallel for schedule(static) num_threads(g->nthreads)
    for (int l = 0; l < g->L; ++l) update_one(g, 0, l);
}

/* out[c] = sum over this shard's factors of the message to camera c: eta 6 | Lambda 36 (dense), no prior */
void gbpo_cam_partial(const gbpo_t *g, double *out)
{
    for (int c = 0; c < g->C; ++c) {
        double *o = out + 42 * (size_t)c;
        for (int k = 0; k < 42; ++k) o[k] = 0.0;
        for (int e = g->cam_ptr[c]; e < g->cam_ptr[c + 1]; ++e) {
            size_t f = (size_t)g->cam_adj[e];
            for (int k = 0; k < 6; ++k) o[k] += g->m_cam_eta[6 * f + k];
            for (int k = 0; k < 36; ++k) o[6 + k] += g->m_cam_lam[36 * f + k];
        }
    }
}

/* belief_c = prior_c + sum over shards (rank order) of their partial; mu_c = inv(Lambda) eta */
void gbpo_cam_finish(gbpo_t *g, const double *gathered, int n_parts)
{
    for (int c = 0; c < g->C; ++c) {
        double eta[6], lam[36], S[36];
        for (int k = 0; k < 6; ++k) eta[k] = g->cam_prior_eta[6 * (size_t)c + k];
        for (int k = 0; k < 36; ++k) lam[k] = g->cam_prior_lam[36 * (size_t)c + k];
        for (int r = 0; r < n_parts; ++r) {
            const double *src = gathered + ((size_t)r * g->C + c) * 42;
            for (int k = 0; k < 6; ++k) eta[k] += src[k];
            for (int k = 0; k < 36; ++k) lam[k] += src[6 + k];
        }
        for (int k = 0; k < 6; ++k) g->cam_bel_eta[6 * (size_t)c + k] = eta[k];
        for (int k = 0; k < 36; ++k) g->cam_bel_lam[36 * (size_t)c + k] = lam[k];
        inv_n(lam, 6, S);
        for (int a = 0; a < 6; ++a) {
            double s = 0.0;
            for (int b = 0; b < 6; ++b) s += S[a * 6 + b] * eta[b];
            g->cam_mu[6 * (size_t)c + a] = s;
        }
    }
}

/* per-variable max over adjacent factors of max(Lambda_f): the first half of generate_priors_var (gbp_ba.py:27-31) */
void gbpo_factor_lambda_max(const gbpo_t *g, double *cam_max, double *lmk_max)
{
    for (int v = 0; v < g->C + g->L; ++v) {
        int is_cam = v < g->C, i = is_cam ? v : v - g->C;
        const int *ptr = is_cam ? g->cam_ptr : g->lmk_ptr, *adj = is_cam ? g->cam_adj : g->lmk_adj;
        double mx = 0.0;
        for (int e = ptr[i]; e < ptr[i + 1]; ++e) {
            const double *lam = g->f_lam + 81 * (size_t)adj[e];
            for (int k = 0; k < 81; ++k) if (lam[k] > mx) mx = lam[k];
        }
        if (is_cam) cam_max[i] = mx; else lmk_max[i] = mx;
    }
}

/* Lambda_prior = lambda I, eta_prior = Lambda_prior mu (gbp_ba.py:32-34) from given per-variable scalars */
void gbpo_set_prior_scalars(gbpo_t *g, const double *cam_lambda, const double *lmk_lambda)
{
    for (int c = 0; c < g->C; ++c)
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) g->cam_prior_lam[36 * (size_t)c + a * 6 + b] = (a == b) ? cam_lambda[c] : 0.0;
            g->cam_prior_eta[6 * (size_t)c + a] = cam_lambda[c] * g->cam_mu[6 * (size_t)c + a];
        }
    for (int l = 0; l < g->L; ++l)
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) g->lmk_prior_lam[9 * (size_t)l + a * 3 + b] = (a == b) ? lmk_lambda[l] : 0.0;
            g->lmk_prior_eta[3 * (size_t)l + a] = lmk_lambda[l] * g->lmk_mu[3 * (size_t)l + a];
        }
}

/* un-normalised diagnostics of one shard: {sum ||r||, sum 0.5 ||r||^2 / var} */
void gbpo_residual_sums(gbpo_t *g, double *out2)
{
    out2[0] = 0.0; out2[1] = 0.0;
    for (int f = 0; f < g->F; ++f) {
        double r[2];
        residual_one(g, f, r);
        double n = sqrt(r[0] * r[0] + r[1] * r[1]);
        out2[0] += n;
        out2[1] += 0.5 * (n * n) / g->adaptive_var[f];
    }
}

/* ------------------------------------------------------------------------- diagnostics -- */

/* Factor.compute_residual: gbp/gbp.py:251-259 */
static void residual_one(const gbpo_t *g, int f, double *r)
{
    double m[9], h[2];
    adj_means(g, f, m);
    meas_fn(m, g->K, h);
    r[0] = h[0] - g->z[2 * (size_t)f];
    r[1] = h[1] - g->z[2 * (size_t)f + 1];
}

/* BAFactorGraph.are: gbp/gbp_ba.py:61-69 */
double gbpo_are(gbpo_t *g)
{
#pragma omp parallel for schedule(static) num_threads(g->nthreads)
    for (int f = 0; f < g->F; ++f) {
        double r[2];
        residual_one(g, f, r);
        g->scratch[f] = sqrt(r[0] * r[0] + r[1] * r[1]);
    }
    double s = 0.0;
    for (int f = 0; f < g->F; ++f) s += g->scratch[f];
    return s / g->F;
}

/* FactorGraph.energy: gbp/gbp.py:36-44 */
double gbpo_energy(gbpo_t *g)
{
#pragma omp parallel for schedule(static) num_threads(g->nthreads)
    for (int f = 0; f < g->F; ++f) {
        double r[2];
        residual_one(g, f, r);
        double n = sqrt(r[0] * r[0] + r[1] * r[1]);
        g->scratch[f] = 0.5 * (n * n) / g->adaptive_var[f];
    }
    double s = 0.0;
    for (int f = 0; f < g->F; ++f) s += g->scratch[f];
    return s;
}

/* ------------------------------------------------------------------- views for the tests -- */

void gbpo_fn_eval(const double *x9, const double *K4, double *h2, double *J18)
{
    double K[9] = {K4[0], 0, K4[2], 0, K4[1], K4[3], 0, 0, 1};
    meas_fn(x9, K, h2);
    jac_fn(x9, K, J18);
}

#define COPY(dst, src, n) do { if (dst) memcpy((dst), (src), sizeof(*(src)) * (size_t)(n)); } while (0)

void gbpo_get_beliefs(const gbpo_t *g, double *ce, double *cl, double *le, double *ll)
{
    COPY(ce, g->cam_bel_eta, 6 * (size_t)g->C); COPY(cl, g->cam_bel_lam, 36 * (size_t)g->C);
    COPY(le, g->lmk_bel_eta, 3 * (size_t)g->L); COPY(ll, g->lmk_bel_lam, 9 * (size_t)g->L);
}
void gbpo_get_means(const gbpo_t *g, double *cm, double *lm)
{
    COPY(cm, g->cam_mu, 6 * (size_t)g->C); COPY(lm, g->lmk_mu, 3 * (size_t)g->L);
}
void gbpo_get_priors(const gbpo_t *g, double *ce, double *cl, double *le, double *ll)
{
    COPY(ce, g->cam_prior_eta, 6 * (size_t)g->C); COPY(cl, g->cam_prior_lam, 36 * (size_t)g->C);
    COPY(le, g->lmk_prior_eta, 3 * (size_t)g->L); COPY(ll, g->lmk_prior_lam, 9 * (size_t)g->L);
}
void gbpo_get_messages(const gbpo_t *g, double *ce, double *cl, double *le, double *ll)
{
    COPY(ce, g->m_cam_eta, 6 * (size_t)g->F); COPY(cl, g->m_cam_lam, 36 * (size_t)g->F);
    COPY(le, g->m_lmk_eta, 3 * (size_t)g->F); COPY(ll, g->m_lmk_lam, 9 * (size_t)g->F);
}
void gbpo_get_factors(const gbpo_t *g, double *eta, double *lam, double *linpoint, int *cam, int *lmk, double *z)
{
    COPY(eta, g->f_eta, 9 * (size_t)g->F); COPY(lam, g->f_lam, 81 * (size_t)g->F);
    COPY(linpoint, g->linpoint, 9 * (size_t)g->F);
    COPY(cam, g->f_cam, g->F); COPY(lmk, g->f_lmk, g->F); COPY(z, g->z, 2 * (size_t)g->F);
}
void gbpo_get_relin_state(const gbpo_t *g, int *iters, double *damping, double *adaptive_var, unsigned char *robust)
{
    COPY(iters, g->iters, g->F); COPY(damping, g->damping, g->F);
    COPY(adaptive_var, g->adaptive_var, g->F); COPY(robust, g->robust, g->F);
}
void gbpo_set_iters_since_relin(gbpo_t *g, const int *iters) { memcpy(g->iters, iters, sizeof(int) * (size_t)g->F); }
void gbpo_fill_iters_since_relin(gbpo_t *g, int v) { for (int f = 0; f < g->F; ++f) g->iters[f] = v; }
