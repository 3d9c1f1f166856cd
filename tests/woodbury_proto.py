"""numpy prototype of the covariance-form ("Woodbury") factor update the HIP sweep uses -- TEST INFRASTRUCTURE.

The reference forms every message by inverting the factor's own block plus the cavity of the variable that is
eliminated (gbp/gbp.py:340-368: a 6x6 inverse for the message to the landmark, a 3x3 for the message to the camera),
once per FACTOR.  With P = Lambda^-1 of the variable's belief (one inverse per VARIABLE) and the message stored as a
2x2 core W in the span of the Jacobian (M = J^T W J, e = J^T q), the same message is 2x2 algebra:

    T = Lambda_belief - J^T W J + s J^T J = Lambda_belief + J^T Q J,   Q = s I - W
    G = J P J^T                                                        (2x2)
    core of the new message  = s I - s^2 J' T^-1 J'^T ... = s (I - W G) (I + Q G)^-1
    coefficients of its eta  = s (I + G Q)^-1 [ rho - J mu + G (q - W rho) ]

(J, W, q, P, mu of the ELIMINATED variable; the result is the message to the OTHER variable).  A factor that
relinearises first takes its old message out of the belief with the old Jacobian (a rank-2 downdate of P and mu) and then
runs the same formulas with W = 0, q = 0 at the new point.

Run as a script: compares the beliefs of this restatement with the C oracle sweep by sweep.
"""
from __future__ import annotations

import numpy as np


def hat(w):
    W = np.zeros(w.shape[:-1] + (3, 3))
    W[..., 0, 1], W[..., 0, 2] = -w[..., 2], w[..., 1]
    W[..., 1, 0], W[..., 1, 2] = w[..., 2], -w[..., 0]
    W[..., 2, 0], W[..., 2, 1] = -w[..., 1], w[..., 0]
    return W


def linearise(x, K):
    """h (F,2), Jc (F,2,6), Jl (F,2,3) of the reprojection factor at x (F,9): reprojection.py:12-44."""
    fx, fy, cx, cy = K
    t, w, y = x[:, 0:3], x[:, 3:6], x[:, 6:9]
    th2 = np.sum(w * w, axis=1)
    th = np.sqrt(th2)
    a = np.sin(th) / th
    b = (1.0 - np.cos(th)) / th2
    Wh = hat(w)
    R = np.eye(3)[None] + a[:, None, None] * Wh + b[:, None, None] * (Wh @ Wh)
    p = np.einsum('fij,fj->fi', R, y) + t
    iz = 1.0 / p[:, 2]
    h = np.stack([(fx * p[:, 0] + cx * p[:, 2]) * iz, (fy * p[:, 1] + cy * p[:, 2]) * iz], axis=1)
    JpK = np.zeros((x.shape[0], 2, 3))
    JpK[:, 0, 0] = fx * iz
    JpK[:, 0, 2] = -fx * p[:, 0] * iz * iz
    JpK[:, 1, 1] = fy * iz
    JpK[:, 1, 2] = -fy * p[:, 1] * iz * iz
    Jl = JpK @ R
    # dR_wx_dw = -R y^ ((R^T - I) w^ + w w^T) / theta^2      derivatives.py:36-45
    inner = (np.transpose(R, (0, 2, 1)) - np.eye(3)[None]) @ Wh + w[:, :, None] * w[:, None, :]
    dR = -(R @ hat(y) @ inner) / th2[:, None, None]
    Jc = np.concatenate([JpK, JpK @ dR], axis=2)
    return h, Jc, Jl


def inv2(M):
    det = M[:, 0, 0] * M[:, 1, 1] - M[:, 0, 1] * M[:, 1, 0]
    out = np.empty_like(M)
    out[:, 0, 0], out[:, 1, 1] = M[:, 1, 1] / det, M[:, 0, 0] / det
    out[:, 0, 1], out[:, 1, 0] = -M[:, 0, 1] / det, -M[:, 1, 0] / det
    return out


I2 = np.eye(2)[None]


def eliminate(J, P, mu, W, q, rho, s):
    """Core and eta coefficients of the message that eliminating the variable (J, P, mu, old message W, q) leaves."""
    G = J @ P @ np.transpose(J, (0, 2, 1))
    Q = s[:, None, None] * I2 - W
    core = s[:, None, None] * ((I2 - W @ G) @ inv2(I2 + Q @ G))
    core = 0.5 * (core + np.transpose(core, (0, 2, 1)))
    rhs = rho - np.einsum('fij,fj->fi', J, mu) + np.einsum('fij,fj->fi', G, q - np.einsum('fij,fj->fi', W, rho))
    r = s[:, None] * np.einsum('fij,fj->fi', inv2(I2 + G @ Q), rhs)
    return core, r


def downdate(J, P, mu, W, q):
    """(P', mu') of belief minus the message (W, q) made with Jacobian J."""
    PJt = P @ np.transpose(J, (0, 2, 1))
    G = J @ PJt
    A = inv2(I2 - W @ G) @ W
    P2 = P + PJt @ A @ np.transpose(PJt, (0, 2, 1))
    t = np.einsum('fij,fj->fi', A, np.einsum('fij,fj->fi', J, mu) - np.einsum('fij,fj->fi', G, q)) - q
    mu2 = mu + np.einsum('fij,fj->fi', PJt, t)
    return P2, mu2


class WoodburyBA:
    def __init__(self, prob, *, gauss_noise_std=2.0, beta=0.01, num_undamped_iters=6, min_linear_iters=8, eta_damping=0.4):
        self.K = np.asarray(prob.K, dtype=np.float64)
        self.cam, self.lmk = prob.cam_idx.astype(np.int64), prob.lmk_idx.astype(np.int64)
        self.C, self.L, self.F = prob.n_cams, prob.n_lmks, prob.n_factors
        self.z = prob.meas.astype(np.float64)
        self.sigma2 = gauss_noise_std ** 2
        self.beta, self.num_undamped, self.min_linear, self.eta_damping = beta, num_undamped_iters, min_linear_iters, eta_damping
        self.x0 = np.concatenate([prob.cam_means[self.cam], prob.lmk_means[self.lmk]], axis=1)
        self.qC, self.qL = np.zeros((self.F, 2)), np.zeros((self.F, 2))
        self.W, self.V = np.zeros((self.F, 2, 2)), np.zeros((self.F, 2, 2))
        self.iters = np.ones(self.F, np.int64)          # gbp.py:249
        self.damped = np.zeros(self.F, bool)
        self.cam_mu, self.lmk_mu = prob.cam_means.copy(), prob.lmk_means.copy()

    def generate_priors_var(self, weaker_factor=100.0):
        h, Jc, Jl = linearise(self.x0, self.K)
        J = np.concatenate([Jc, Jl], axis=2)
        lam = np.einsum('fri,frj->fij', J, J) / self.sigma2
        fmax = lam.reshape(self.F, -1).max(axis=1)
        cmax, lmax = np.zeros(self.C), np.zeros(self.L)
        np.maximum.at(cmax, self.cam, fmax)
        np.maximum.at(lmax, self.lmk, fmax)
        self.cpri_lam = np.eye(6)[None] * (cmax / weaker_factor ** 2)[:, None, None]
        self.lpri_lam = np.eye(3)[None] * (lmax / weaker_factor ** 2)[:, None, None]
        self.cpri_eta = np.einsum('cij,cj->ci', self.cpri_lam, self.cam_mu)
        self.lpri_eta = np.einsum('cij,cj->ci', self.lpri_lam, self.lmk_mu)

    def update_all_beliefs(self):
        h, Jc, Jl = linearise(self.x0, self.K)
        eC = np.einsum('fri,fr->fi', Jc, self.qC)
        MC = np.einsum('fri,frs,fsj->fij', Jc, self.W, Jc)
        eL = np.einsum('fri,fr->fi', Jl, self.qL)
        ML = np.einsum('fri,frs,fsj->fij', Jl, self.V, Jl)
        self.cam_eta, self.cam_lam = self.cpri_eta.copy(), self.cpri_lam.copy()
        self.lmk_eta, self.lmk_lam = self.lpri_eta.copy(), self.lpri_lam.copy()
        np.add.at(self.cam_eta, self.cam, eC)
        np.add.at(self.cam_lam, self.cam, MC)
        np.add.at(self.lmk_eta, self.lmk, eL)
        np.add.at(self.lmk_lam, self.lmk, ML)
        self.cam_P, self.lmk_P = np.linalg.inv(self.cam_lam), np.linalg.inv(self.lmk_lam)
        self.cam_mu = np.einsum('cij,cj->ci', self.cam_P, self.cam_eta)
        self.lmk_mu = np.einsum('cij,cj->ci', self.lmk_P, self.lmk_eta)

    def synchronous_iteration(self):
        F = self.F
        s = np.full(F, 1.0 / self.sigma2)
        mu = np.concatenate([self.cam_mu[self.cam], self.lmk_mu[self.lmk]], axis=1)
        dist = np.linalg.norm(self.x0 - mu, axis=1)
        relin = (dist > self.beta) & (self.iters >= self.min_linear)
        self.iters = np.where(relin, 0, self.iters + 1)
        self.damped = np.where(relin, False, self.damped) | (self.iters == self.num_undamped)
        d = np.where(self.damped, self.eta_damping, 0.0)[:, None]
        Pc, muc, Pl, mul = self.cam_P[self.cam], self.cam_mu[self.cam], self.lmk_P[self.lmk], self.lmk_mu[self.lmk]
        W, V, qC, qL = self.W.copy(), self.V.copy(), self.qC.copy(), self.qL.copy()
        if relin.any():
            i = np.nonzero(relin)[0]
            _, Jco, Jlo = linearise(self.x0[i], self.K)
            Pc[i], muc[i] = downdate(Jco, Pc[i], muc[i], W[i], qC[i])
            Pl[i], mul[i] = downdate(Jlo, Pl[i], mul[i], V[i], qL[i])
            W[i], V[i], qC[i], qL[i] = 0.0, 0.0, 0.0, 0.0
            self.x0[i] = mu[i]
        h, Jc, Jl = linearise(self.x0, self.K)
        rho = np.einsum('fri,fi->fr', Jc, self.x0[:, :6]) + np.einsum('fri,fi->fr', Jl, self.x0[:, 6:]) + self.z - h
        Vn, rL = eliminate(Jc, Pc, muc, W, qC, rho, s)
        Wn, rC = eliminate(Jl, Pl, mul, V, qL, rho, s)
        self.qL = (1.0 - d) * rL + d * self.qL
        self.qC = (1.0 - d) * rC + d * self.qC
        self.V, self.W = Vn, Wn
        self.update_all_beliefs()
        return int(relin.sum())


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gbp_amd import balio, synthetic
    from oracle.oracle import OracleBA
    which = sys.argv[1] if len(sys.argv) > 1 else 'fr1desk_small'
    if which == 'synth':
        prob = synthetic.make_problem(n_cams=20, n_lmks=400, obs_per_lmk=5, seed=1)
    else:
        prob = balio.read_bal(os.path.join(os.path.dirname(__file__), 'golden', 'data', which + '.txt'), native=False)
    o = OracleBA.from_problem(prob)
    o.generate_priors_var()
    o.update_all_beliefs()
    w = WoodburyBA(prob)
    w.generate_priors_var()
    w.update_all_beliefs()
    for it in range(30):
        if it in (3, 8):
            o.set_iters_since_relin(1)
            w.iters[:] = 1
        o.synchronous_iteration(robustify=True, local_relin=True)
        n = w.synchronous_iteration()
        ce, cl, le, ll = o.beliefs()
        rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
        print(f"sweep {it + 1:2d} relin {n:5d}  cam eta {rel(w.cam_eta, ce):.2e} lam {rel(w.cam_lam, cl):.2e}   lmk eta {rel(w.lmk_eta, le):.2e} lam {rel(w.lmk_lam, ll):.2e}")


if __name__ == '__main__':
    main()
