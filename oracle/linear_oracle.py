"""CPU oracle of the LINEAR pairwise GBP path (TEST INFRASTRUCTURE ONLY -- imported by tests/, never by gbp_amd/).

Dense numpy restatement of what joeaortiz/gbp does for a FactorGraph(nonlinear_factors=False) of two-variable
factors (ndim_posegraph.py): no packing, no elimination tricks, np.linalg.inv exactly where the reference has it.
Pinned by tests/test_linear_oracle.py against fixture G8 (the reference's own ndim_posegraph.py run).

  compute_messages   gbp/gbp.py:334-373     update_belief   gbp/gbp.py:176-198     energy   gbp/gbp.py:36-44, 251-265
  synchronous_iteration (linear graph: no robustify, no relinearisation, graph damping)   gbp/gbp.py:46-58, 86-92
"""
import numpy as np


class LinearOracle:
    def __init__(self, var_a, var_b, factor_eta, factor_lam, prior_eta, prior_lam, factor_const=None, eta_damping=0.0):
        self.va, self.vb = np.asarray(var_a, dtype=int), np.asarray(var_b, dtype=int)
        self.fe, self.fl = np.asarray(factor_eta, dtype=float), np.asarray(factor_lam, dtype=float)
        self.pe, self.pl = np.asarray(prior_eta, dtype=float), np.asarray(prior_lam, dtype=float)
        self.N, self.D = self.pe.shape
        self.F = self.va.shape[0]
        self.fc = np.zeros(self.F) if factor_const is None else np.asarray(factor_const, dtype=float)
        self.damping = float(eta_damping)
        D = self.D
        self.msg_eta = np.zeros((self.F, 2, D))            # Factor.messages[k].eta / .lam, zero at construction (gbp.py:222)
        self.msg_lam = np.zeros((self.F, 2, D, D))
        self.bel_eta, self.bel_lam, self.mu = np.zeros((self.N, D)), np.zeros((self.N, D, D)), np.zeros((self.N, D))
        self.adj = [[] for _ in range(self.N)]             # (factor, side) in ascending factor id = append order
        for f in range(self.F):
            self.adj[self.va[f]].append((f, 0))
            self.adj[self.vb[f]].append((f, 1))

    def update_all_beliefs(self):                          # gbp.py:56-58 -> 176-198
        for v in range(self.N):
            eta, lam = self.pe[v].copy(), self.pl[v].copy()
            for f, side in self.adj[v]:
                eta = eta + self.msg_eta[f, side]
                lam = lam + self.msg_lam[f, side]
            self.bel_eta[v], self.bel_lam[v] = eta, lam
            self.mu[v] = np.linalg.inv(lam) @ eta

    def compute_all_messages(self):                        # gbp.py:46-54 (graph-level damping), 334-373
        D = self.D
        new_eta, new_lam = np.empty_like(self.msg_eta), np.empty_like(self.msg_lam)
        for f in range(self.F):
            vs = (self.va[f], self.vb[f])
            for out in (0, 1):
                oth = 1 - out
                eta, lam = self.fe[f].copy(), self.fl[f].copy()
                s = slice(oth * D, (oth + 1) * D)
                eta[s] += self.bel_eta[vs[oth]] - self.msg_eta[f, oth]
                lam[s, s] += self.bel_lam[vs[oth]] - self.msg_lam[f, oth]
                o = slice(out * D, (out + 1) * D)
                gain = lam[o, s] @ np.linalg.inv(lam[s, s])
                new_lam[f, out] = lam[o, o] - gain @ lam[s, o]
                new_eta[f, out] = (1 - self.damping) * (eta[o] - gain @ eta[s]) + self.damping * self.msg_eta[f, out]
        self.msg_eta, self.msg_lam = new_eta, new_lam

    def synchronous_iteration(self):                       # gbp.py:86-92
        self.compute_all_messages()
        self.update_all_beliefs()

    def iterate(self, n):
        for _ in range(n):
            self.synchronous_iteration()

    def energy(self):                                      # 0.5 |h(mu) - z|^2 / sigma^2 written through (eta_f, Lambda_f, const)
        e = 0.0
        for f in range(self.F):
            x = np.concatenate([self.mu[self.va[f]], self.mu[self.vb[f]]])
            e += 0.5 * x @ self.fl[f] @ x - self.fe[f] @ x + self.fc[f]
        return e

    def get_means(self):
        return self.mu.reshape(-1).copy()

    def beliefs(self):
        return self.bel_eta.copy(), self.bel_lam.copy()

    def messages(self):
        return self.msg_eta[:, 0].copy(), self.msg_lam[:, 0].copy(), self.msg_eta[:, 1].copy(), self.msg_lam[:, 1].copy()


def toy_posegraph(n=100, dim=3, M=10, std=1.0, seed=0):
    """The graph of ndim_posegraph.py:36-64 as arrays: (var_a, var_b, factor_eta, factor_lam, factor_const, prior_eta, prior_lam).
    Factor = linear_displacement (gbp/factors/linear_displacement.py:8-14): h(x) = x_b - x_a, J = [-I, I]."""
    rs = np.random.RandomState(seed)
    priors_mu = rs.rand(n, dim) * 10
    prior_lam = np.linalg.inv(3 * np.eye(dim))
    pairs, meas = [], []
    for i, mu in enumerate(priors_mu):
        d = np.array([np.linalg.norm(mu - m1) for m1 in priors_mu])
        for j in d.argsort()[1:M + 1]:
            if [j, i] not in pairs:
                meas.append(mu - priors_mu[j] + rs.normal(0., std, dim))
                pairs.append([i, j])
    J = np.hstack([-np.eye(dim), np.eye(dim)])
    fe = np.array([J.T @ z / std ** 2 for z in meas])
    fl = np.array([J.T @ J / std ** 2 for _ in meas])
    fc = np.array([0.5 * z @ z / std ** 2 for z in meas])
    pairs = np.array(pairs)
    return (pairs[:, 0], pairs[:, 1], fe, fl, fc, priors_mu @ prior_lam.T, np.tile(prior_lam, (n, 1, 1)))
