"""Generic Gaussian belief propagation on the host (numpy): the reference's `gbp.gbp` module surface.

FactorGraph / VariableNode / Factor accept arbitrary variable sizes and arbitrary Python measurement and Jacobian
callables, which is what ndim_posegraph.py needs (BASELINE.json config 1: "plumbing, no GPU").  Bundle adjustment does
not go through these classes: `gbp.gbp_ba` runs on the MI355X.  Semantics follow joeaortiz/gbp gbp/gbp.py (cited per
method); the code is written against that behaviour, not copied from it.
"""
import numpy as np

from utils.gaussian import NdimGaussian


def _belief_mean(g):
    return np.linalg.solve(g.lam, g.eta)


class FactorGraph:
    """gbp/gbp.py:11-153."""

    def __init__(self, nonlinear_factors=True, eta_damping=0.0, beta=None, num_undamped_iters=None, min_linear_iters=None):
        self.var_nodes, self.factors = [], []
        self.n_var_nodes = self.n_factor_nodes = self.n_edges = 0
        self.nonlinear_factors = nonlinear_factors
        self.eta_damping = eta_damping
        if nonlinear_factors:
            self.beta = beta                              # relinearise when the belief means moved further than this
            self.num_undamped_iters = num_undamped_iters  # undamped sweeps after a relinearisation
            self.min_linear_iters = min_linear_iters      # sweeps a factor must stay linear

    # -- one sweep (gbp.py:86-92) ---------------------------------------------------------------------------------
    def synchronous_iteration(self, local_relin=True, robustify=False):
        if robustify:
            self.robustify_all_factors()
        if self.nonlinear_factors and local_relin:
            self.relinearise_factors()
        self.compute_all_messages(local_relin=local_relin)
        self.update_all_beliefs()

    def robustify_all_factors(self):
        for f in self.factors:
            f.robustify_loss()

    def relinearise_factors(self):
        """gbp.py:64-80: per factor, relinearise at the belief means when they drifted more than beta."""
        if not self.nonlinear_factors:
            return
        for f in self.factors:
            means = np.concatenate([_belief_mean(b) for b in f.adj_beliefs])
            if np.linalg.norm(np.asarray(f.linpoint) - means) > self.beta and f.iters_since_relin >= self.min_linear_iters:
                f.compute_factor(linpoint=means)
                f.iters_since_relin = 0
                f.eta_damping = 0.0
            else:
                f.iters_since_relin += 1

    def compute_all_messages(self, local_relin=True):
        """gbp.py:46-54: per-factor damping when relinearisation is local, graph damping otherwise."""
        per_factor = self.nonlinear_factors and local_relin
        for f in self.factors:
            if per_factor:
                if f.iters_since_relin == self.num_undamped_iters:
                    f.eta_damping = self.eta_damping
                f.compute_messages(f.eta_damping)
            else:
                f.compute_messages(self.eta_damping)

    def update_all_beliefs(self):
        for v in self.var_nodes:
            v.update_belief()

    def compute_all_factors(self):
        for f in self.factors:
            f.compute_factor()

    # -- diagnostics / batch solution -----------------------------------------------------------------------------
    def energy(self):
        """gbp.py:36-44."""
        return sum(f.energy() for f in self.factors)

    def get_means(self):
        return np.concatenate([np.asarray(v.mu, dtype=float) for v in self.var_nodes]) if self.var_nodes else np.array([])

    def joint_distribution_inf(self):
        """Joint (eta, Lambda) over all variables at the current linearisation (gbp.py:94-134)."""
        offsets, n = {}, 0
        for v in self.var_nodes:
            offsets[v.variableID] = n
            n += v.dofs
        eta, lam = np.zeros(n), np.zeros((n, n))
        for v in self.var_nodes:
            o = offsets[v.variableID]
            eta[o:o + v.dofs] += v.prior.eta
            lam[o:o + v.dofs, o:o + v.dofs] += v.prior.lam
        for f in self.factors:
            spans, start = [], 0
            for v in f.adj_var_nodes:
                spans.append((offsets[v.variableID], start, v.dofs))
                start += v.dofs
            for go, fo, d in spans:
                eta[go:go + d] += f.factor.eta[fo:fo + d]
                for go2, fo2, d2 in spans:
                    lam[go:go + d, go2:go2 + d2] += f.factor.lam[fo:fo + d, fo2:fo2 + d2]
        return eta, lam

    def device_engine(self, device=0):
        """No reference counterpart: this LINEAR graph (nonlinear_factors=False, two-variable factors, d <= 6) on the MI355X
        (include/gbp_lin.h).  Call after compute_all_factors(); the host graph is left untouched."""
        from gbp_amd.linear import LinearEngine
        return LinearEngine.from_factor_graph(self, device=device)

    def joint_distribution_cov(self):
        eta, lam = self.joint_distribution_inf()
        sigma = np.linalg.inv(lam)
        return sigma @ eta, sigma


class VariableNode:
    """gbp/gbp.py:156-198."""

    def __init__(self, variable_id, dofs):
        self.variableID = variable_id
        self.dofs = dofs
        self.adj_factors = []
        self.mu = np.zeros(dofs)
        self.Sigma = np.zeros((dofs, dofs))
        self.belief = NdimGaussian(dofs)
        self.prior = NdimGaussian(dofs)
        self.prior_lambda_end = -1
        self.prior_lambda_logdiff = -1

    def update_belief(self):
        """belief = prior x incoming messages (in adj_factors order); then hand the belief to the adjacent factors."""
        eta, lam = np.array(self.prior.eta, dtype=float), np.array(self.prior.lam, dtype=float)
        slots = []
        for f in self.adj_factors:
            k = f.adj_vIDs.index(self.variableID)
            slots.append((f, k))
            eta = eta + f.messages[k].eta
            lam = lam + f.messages[k].lam
        self.belief.eta, self.belief.lam = eta, lam
        self.Sigma = np.linalg.inv(lam)
        self.mu = self.Sigma @ eta
        for f, k in slots:
            f.adj_beliefs[k].eta, f.adj_beliefs[k].lam = eta, lam


class Factor:
    """gbp/gbp.py:201-373.  Extra positional arguments are forwarded to meas_fn / jac_fn."""

    def __init__(self, factor_id, adj_var_nodes, measurement, gauss_noise_std, meas_fn, jac_fn, loss=None,
                 mahalanobis_threshold=2, *args):
        self.factorID = factor_id
        self.adj_var_nodes = adj_var_nodes
        self.adj_vIDs = [v.variableID for v in adj_var_nodes]
        self.adj_beliefs = [NdimGaussian(v.dofs) for v in adj_var_nodes]
        self.messages = [NdimGaussian(v.dofs) for v in adj_var_nodes]
        self.dofs_conditional_vars = sum(v.dofs for v in adj_var_nodes)
        self.factor = NdimGaussian(self.dofs_conditional_vars)
        self.linpoint = np.zeros(self.dofs_conditional_vars)
        self.measurement = measurement
        self.gauss_noise_var = gauss_noise_std ** 2
        self.adaptive_gauss_noise_var = gauss_noise_std ** 2
        self.meas_fn, self.jac_fn, self.args = meas_fn, jac_fn, args
        self.loss = loss
        self.mahalanobis_threshold = mahalanobis_threshold
        self.robust_flag = False
        self.eta_damping = 0.0
        self.iters_since_relin = 1

    def _belief_means(self):
        return np.concatenate([_belief_mean(b) for b in self.adj_beliefs])

    def compute_residual(self):
        """h(belief means) - z   (gbp.py:251-259)."""
        return self.meas_fn(self._belief_means(), *self.args) - self.measurement

    def energy(self):
        r = self.compute_residual()
        return 0.5 * float(np.dot(np.atleast_1d(r), np.atleast_1d(r))) / self.adaptive_gauss_noise_var

    def compute_factor(self, linpoint=None, update_self=True):
        """Linearise at `linpoint` (default: adjacent belief means): Lambda = J^T J / var, eta = J^T (J x0 + z - h) / var."""
        if linpoint is None:
            self.linpoint = [float(x) for x in self._belief_means()]     # the reference keeps a list here (gbp.py:274-276)
        else:
            self.linpoint = linpoint
        x0 = np.asarray(self.linpoint, dtype=float)
        J = np.atleast_2d(self.jac_fn(self.linpoint, *self.args))
        innovation = J @ x0 + self.measurement - self.meas_fn(self.linpoint, *self.args)
        w = 1.0 / self.adaptive_gauss_noise_var
        lam = w * (J.T @ J)
        eta = w * (J.T @ np.atleast_1d(innovation))
        if update_self:
            self.factor.eta, self.factor.lam = eta, lam
        return eta, lam

    def robustify_loss(self):
        """Adaptive noise variance from the residual at the LINEARISATION POINT (gbp.py:296-332)."""
        old = self.adaptive_gauss_noise_var
        new = self.gauss_noise_var
        if self.loss is not None:
            r = np.atleast_1d(self.measurement - self.meas_fn(self.linpoint, *self.args))
            m = float(np.sqrt(np.dot(r, r))) / np.sqrt(self.gauss_noise_var)
            self.robust_flag = bool(m > self.mahalanobis_threshold)
            if self.robust_flag and self.loss == 'huber':
                t = self.mahalanobis_threshold
                new = self.gauss_noise_var * m ** 2 / (2 * (t * m - 0.5 * t ** 2))
            elif self.robust_flag and self.loss == 'constant':
                new = m ** 2                                             # as in the reference (no sigma^2 factor, gbp.py:324)
            elif self.loss not in ('huber', 'constant'):
                new = old
        self.adaptive_gauss_noise_var = new
        self.factor.eta = self.factor.eta * (old / new)
        self.factor.lam = self.factor.lam * (old / new)

    def compute_messages(self, eta_damping):
        """All outgoing messages from the OLD incoming ones, committed together (gbp.py:334-373)."""
        sizes = [v.dofs for v in self.adj_var_nodes]
        starts = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
        total = int(starts[-1])
        new = []
        for out, (o0, o1) in enumerate(zip(starts[:-1], starts[1:])):
            eta = np.array(self.factor.eta, dtype=float)
            lam = np.array(self.factor.lam, dtype=float)
            for k, (s0, s1) in enumerate(zip(starts[:-1], starts[1:])):
                if k != out:                      # fold in the other variables' beliefs minus what we told them
                    eta[s0:s1] += self.adj_beliefs[k].eta - self.messages[k].eta
                    lam[s0:s1, s0:s1] += self.adj_beliefs[k].lam - self.messages[k].lam
            keep = np.arange(o0, o1)
            drop = np.concatenate([np.arange(0, o0), np.arange(o1, total)]).astype(int)
            gain = lam[np.ix_(keep, drop)] @ np.linalg.inv(lam[np.ix_(drop, drop)])
            msg_lam = lam[np.ix_(keep, keep)] - gain @ lam[np.ix_(drop, keep)]
            msg_eta = eta[keep] - gain @ eta[drop]
            new.append(((1 - eta_damping) * msg_eta + eta_damping * self.messages[out].eta, msg_lam))
        for k, (e, l) in enumerate(new):
            self.messages[k].eta, self.messages[k].lam = e, l
