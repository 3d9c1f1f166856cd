"""The reference's BA graph on the HOST, object per factor / numpy per operation -- TEST INFRASTRUCTURE and the second CPU
baseline of bench.py ("the build's numpy restatement at 1 thread on configs 2-3: the closest analogue of the
reference's cost model").  It assembles what create_ba_graph (gbp/gbp_ba.py:97-150) assembles, from the generic
FactorGraph / VariableNode / Factor classes of the drop-in `gbp.gbp` module (gbp_amd/compat/gbp/gbp.py, the numpy host path
that runs ndim_posegraph.py) and its reprojection meas_fn / jac_fn, then sweeps it like ba.py does.  Pinned to fixture G4
in tests/test_oracle_golden.py.  Nothing in gbp_amd/ imports this module.
"""
import os
import sys

import numpy as np

_COMPAT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gbp_amd', 'compat')


def _compat_modules():
    """Import the drop-in `gbp` / `utils` packages from gbp_amd/compat without disturbing whatever carries those names already
    (the reference itself in the golden generator, nothing in the tests)."""
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in ('gbp', 'utils', 'vis')}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, _COMPAT)
    try:
        from gbp import gbp
        from gbp.factors import reprojection
    finally:
        sys.path.remove(_COMPAT)
        for k in list(sys.modules):
            if k.split('.')[0] in ('gbp', 'utils', 'vis'):
                del sys.modules[k]
        sys.modules.update(saved)
    return gbp, reprojection


class NumpyBA:
    """BAFactorGraph surface (generate_priors_var / update_all_beliefs / synchronous_iteration / are / energy / beliefs)."""

    def __init__(self, problem, gauss_noise_std=2.0, loss=None, Nstds=3.0, beta=0.01, num_undamped_iters=6, min_linear_iters=8,
                 eta_damping=0.4):
        gbp, reprojection = _compat_modules()
        p = problem
        K = np.array([[p.K[0], 0.0, p.K[2]], [0.0, p.K[1], p.K[3]], [0.0, 0.0, 1.0]])
        g = gbp.FactorGraph(nonlinear_factors=True, eta_damping=eta_damping, beta=beta, num_undamped_iters=num_undamped_iters,
                            min_linear_iters=min_linear_iters)
        self.C, self.L = p.n_cams, p.n_lmks
        cams, lmks = [], []
        for c in range(p.n_cams):
            v = gbp.VariableNode(c, 6)
            v.mu = np.array(p.cam_means[c])
            cams.append(v)
        for l in range(p.n_lmks):
            v = gbp.VariableNode(p.n_cams + l, 3)
            v.mu = np.array(p.lmk_means[l])
            lmks.append(v)
        order = np.argsort(p.cam_idx, kind='stable')            # camera-major, file order inside a camera (gbp_ba.py:128-130)
        for fid, i in enumerate(order):
            cv, lv = cams[p.cam_idx[i]], lmks[p.lmk_idx[i]]
            f = gbp.Factor(fid, [cv, lv], np.array(p.meas[i]), gauss_noise_std, reprojection.meas_fn, reprojection.jac_fn, loss, Nstds, K)
            f.compute_factor(linpoint=np.concatenate([cv.mu, lv.mu]))
            cv.adj_factors.append(f)
            lv.adj_factors.append(f)
            g.factors.append(f)
        g.var_nodes = cams + lmks
        g.n_var_nodes, g.n_factor_nodes, g.n_edges = len(g.var_nodes), len(g.factors), 2 * len(g.factors)
        self.graph, self.cams, self.lmks = g, cams, lmks

    def generate_priors_var(self, weaker_factor=100.0):
        """gbp_ba.py:20-34: prior Lambda = I max_f max(Lambda_f) / w^2, eta = Lambda mu."""
        for v in self.graph.var_nodes:
            m = 0.0
            for f in v.adj_factors:
                m = max(m, float(np.max(f.factor.lam)))
            lam = np.eye(v.dofs) * m / (weaker_factor ** 2)
            v.prior.lam, v.prior.eta = lam, lam @ v.mu

    def weaken_priors(self, factor):
        for v in self.graph.var_nodes:
            v.prior.eta, v.prior.lam = v.prior.eta * factor, v.prior.lam * factor

    def update_all_beliefs(self):
        self.graph.update_all_beliefs()

    def synchronous_iteration(self, local_relin=True, robustify=False):
        self.graph.synchronous_iteration(local_relin=local_relin, robustify=robustify)

    def iterate(self, n, robustify=True, local_relin=True):
        for _ in range(int(n)):
            self.graph.synchronous_iteration(local_relin=local_relin, robustify=robustify)

    def set_iters_since_relin(self, v):
        for f in self.graph.factors:
            f.iters_since_relin = int(v)

    def are(self):
        """gbp_ba.py:61-69."""
        return float(sum(np.linalg.norm(f.compute_residual()) for f in self.graph.factors) / len(self.graph.factors))

    def energy(self):
        return float(self.graph.energy())

    def beliefs(self):
        return (np.array([v.belief.eta for v in self.cams]), np.array([v.belief.lam for v in self.cams]),
                np.array([v.belief.eta for v in self.lmks]), np.array([v.belief.lam for v in self.lmks]))

    def relin_state(self):
        return dict(iters_since_relin=np.array([f.iters_since_relin for f in self.graph.factors], dtype=np.int32))
