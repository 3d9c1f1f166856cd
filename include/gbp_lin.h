/* gbp_lin.h -- C ABI of the MI355X engine for LINEAR pairwise Gaussian belief propagation (libgbp_hip.so).
 *
 * the generic path of joeaortiz/gbp (gbp/gbp.py FactorGraph with
 * nonlinear_factors=False, as built by ndim_posegraph.py with gbp/factors/linear_displacement.py:8-14)
 * for graphs whose factors all join TWO variables of the same size d <= 6.  A linear factor never
 * relinearises, so it is handed over once as its information form (eta_f, Lambda_f) over the
 * stacked variables [a; b] (Factor.compute_factor gbp.py:267-294 evaluated by the caller), and the
 * device runs FactorGraph.synchronous_iteration (gbp.py:86-92) = compute_all_messages with the
 * graph-level damping (gbp.py:52-54) + update_all_beliefs (gbp.py:56-58).
 *
 * Conventions as in gbp_ba.h: 0 or a negative GBP_E* code, message from gbp_last_error(); host
 * pointers caller-owned, contiguous C-order float64 / int32; dense matrices row-major; information form.
 * Variables and factors keep the caller's numbering; a variable's adjacency order (the order its
 * belief adds messages in, VariableNode.adj_factors gbp.py:160) is ascending factor id, which is what
 * ndim_posegraph.py:86-88 produces.
 */
#ifndef GBP_LIN_H
#define GBP_LIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBP_LIN_MAX_DOFS 6

typedef struct gbp_lin gbp_lin_t;

typedef struct {
    int32_t n_vars, dofs, n_factors, device;
    const int32_t *var_a, *var_b;  /* F each: Factor.adj_vIDs (gbp.py:225), a != b                          */
    const double *factor_eta;      /* F x 2d            Factor.factor.eta  gbp.py:291                         */
    const double *factor_lam;      /* F x 2d x 2d       Factor.factor.lam  gbp.py:292                         */
    const double *factor_const;    /* F or NULL: 0.5 |z - h(0)|^2 / sigma^2, the constant of Factor.energy gbp.py:261-265 */
    const double *prior_eta;       /* N x d             VariableNode.prior.eta  gbp.py:170                    */
    const double *prior_lam;       /* N x d x d         VariableNode.prior.lam                                 */
    double eta_damping;            /* FactorGraph.eta_damping  gbp.py:18                                       */
} gbp_lin_desc_t;

int  gbp_lin_create(gbp_lin_t **out, const gbp_lin_desc_t *d);                  /* graph construction ndim_posegraph.py:67-91 */
void gbp_lin_destroy(gbp_lin_t *h);
int  gbp_lin_sync(gbp_lin_t *h);
int  gbp_lin_update_beliefs(gbp_lin_t *h);                                      /* FactorGraph.update_all_beliefs gbp.py:56-58 */
int  gbp_lin_iterate(gbp_lin_t *h, int32_t n_iters);                            /* n x synchronous_iteration gbp.py:86-92      */
int  gbp_lin_energy(gbp_lin_t *h, double *out);                                 /* FactorGraph.energy gbp.py:36-44             */
int  gbp_lin_get_beliefs(gbp_lin_t *h, double *eta, double *lam);               /* N x d, N x d x d   VariableNode.belief      */
int  gbp_lin_get_means(gbp_lin_t *h, double *mu);                               /* N x d   FactorGraph.get_means gbp.py:146-153 */
int  gbp_lin_get_messages(gbp_lin_t *h, double *eta_a, double *lam_a, double *eta_b, double *lam_b);   /* Factor.messages gbp.py:222 */

#ifdef __cplusplus
}
#endif
#endif /* GBP_LIN_H */
