"""The sweep kernels' per-factor C++ (gbp_math.hpp + factor_core of gbp_kernels.hpp) compiled for the HOST and driven factor by
factor through whole sweeps, against the C oracle (dense reference maths).  No GPU: this pins the covariance-form algebra,
the relinearisation downdate, robust losses and the dense-remainder path of the very code the GPU runs.  Test infrastructure only
(tests/hostmath/host_math.hip); the product has no CPU path."""
import ctypes as ct
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import DATA, rel_err_rows
from woodbury_proto import linearise

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'hostmath', 'host_math.hip')
LIB = os.path.join(HERE, 'hostmath', 'libhostmath.so')
CSRC = os.path.join(os.path.dirname(HERE), 'gbp_amd', 'csrc')

IU6, IU3 = np.triu_indices(6), np.triu_indices(3)
_dp, _ip = ct.POINTER(ct.c_double), ct.POINTER(ct.c_int)


@pytest.fixture(scope='module')
def hm():
    if os.environ.get('GBP_HOSTMATH_SANITIZED_LIB'):          # tests/test_sanitizers.py: the same source under ASan + UBSan
        return ct.CDLL(os.environ['GBP_HOSTMATH_SANITIZED_LIB'])
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    deps = [SRC] + [os.path.join(CSRC, f) for f in ('gbp_math.hpp', 'gbp_kernels.hpp')]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=fast', '-o', LIB, SRC])
    return ct.CDLL(LIB)


def d(a):
    return a.ctypes.data_as(_dp)


def unpack(pk, n):
    iu = IU6 if n == 6 else IU3
    out = np.zeros((pk.shape[0], n, n))
    out[:, iu[0], iu[1]] = pk
    out[:, iu[1], iu[0]] = pk
    return out


class HostBA:
    """BAFactorGraph life cycle on the host-compiled kernel maths (sums and bookkeeping in numpy)."""

    def __init__(self, lib, prob, *, gauss_noise_std=2.0, loss=None, Nstds=3.0, beta=0.01, num_undamped_iters=6, min_linear_iters=8,
                 eta_damping=0.4):
        self.lib = lib
        self.K = np.ascontiguousarray(prob.K, dtype=np.float64)
        order = np.argsort(prob.cam_idx, kind='stable')          # reference factor order (gbp_ba.py:128-130)
        self.cam, self.lmk = prob.cam_idx[order].astype(np.int64), prob.lmk_idx[order].astype(np.int64)
        self.C, self.L, self.F = prob.n_cams, prob.n_lmks, prob.n_factors
        self.z = np.ascontiguousarray(prob.meas[order], dtype=np.float64)
        self.sigma2 = gauss_noise_std ** 2
        self.par = dict(nstds=Nstds, beta=beta, eta_damping=eta_damping, num_undamped=num_undamped_iters, min_linear=min_linear_iters,
                        loss={None: 0, 'huber': 1, 'constant': 2}[loss])
        F = self.F
        self.x0 = np.ascontiguousarray(np.concatenate([prob.cam_means[self.cam], prob.lmk_means[self.lmk]], axis=1))
        self.clk = 0                                                             # the graph's relinearisation clock (gbp_kernels.hpp, state word)
        self.st = np.full(F, lib.hm_state_pack(1, 0, 0, 0, 0, 0), np.int32)    # iters_since_relin = 1, gbp.py:249
        self.avar = np.full(F, self.sigma2)
        self.qC, self.qL, self.WC, self.VL = np.zeros((F, 2)), np.zeros((F, 2)), np.zeros((F, 3)), np.zeros((F, 3))
        self.xt = np.zeros((F, 9)) if num_undamped_iters == 0 else None
        self.eC, self.MC, self.eL, self.ML = np.zeros((F, 6)), np.zeros((F, 21)), np.zeros((F, 3)), np.zeros((F, 6))
        self.cam_mu, self.lmk_mu = prob.cam_means.astype(np.float64).copy(), prob.lmk_means.astype(np.float64).copy()

    def generate_priors_var(self, wf=100.0):
        _, Jc, Jl = linearise(self.x0, self.K)
        J = np.concatenate([Jc, Jl], axis=2)
        fmax = (np.einsum('fri,frj->fij', J, J) / self.avar[:, None, None]).reshape(self.F, -1).max(axis=1)
        cmax, lmax = np.zeros(self.C), np.zeros(self.L)
        np.maximum.at(cmax, self.cam, fmax)
        np.maximum.at(lmax, self.lmk, fmax)
        self.cpri = np.zeros((self.C, 27))
        self.lpri = np.zeros((self.L, 9))
        for k in range(6):
            self.cpri[:, 6 + [0, 6, 11, 15, 18, 20][k]] = cmax / wf ** 2
        for k in range(3):
            self.lpri[:, 3 + [0, 3, 5][k]] = lmax / wf ** 2
        self.cpri[:, :6] = (cmax / wf ** 2)[:, None] * self.cam_mu
        self.lpri[:, :3] = (lmax / wf ** 2)[:, None] * self.lmk_mu

    def weaken_priors(self, f):
        self.cpri *= f
        self.lpri *= f

    def update_all_beliefs(self):
        cb, lb = self.cpri.copy(), self.lpri.copy()
        np.add.at(cb, self.cam, np.concatenate([self.eC, self.MC], axis=1))
        np.add.at(lb, self.lmk, np.concatenate([self.eL, self.ML], axis=1))
        self.cbel, self.lbel = cb, lb
        self.cam_mu, self.cam_P = np.empty((self.C, 6)), np.empty((self.C, 21))
        self.lmk_mu, self.lmk_P = np.empty((self.L, 3)), np.empty((self.L, 6))
        self.lib.hm_belief(6, self.C, d(np.ascontiguousarray(cb[:, :6])), d(np.ascontiguousarray(cb[:, 6:])), d(self.cam_mu), d(self.cam_P))
        self.lib.hm_belief(3, self.L, d(np.ascontiguousarray(lb[:, :3])), d(np.ascontiguousarray(lb[:, 3:])), d(self.lmk_mu), d(self.lmk_P))

    def set_iters_since_relin(self, v):
        rc = (self.clk - int(v)) & 0xfffff
        self.st = ((self.st.astype(np.int64) & 0xfff) | (rc << 12)).astype(np.uint32).view(np.int32)

    def synchronous_iteration(self, robustify=True, local_relin=True):
        muC, PC = np.ascontiguousarray(self.cam_mu[self.cam]), np.ascontiguousarray(self.cam_P[self.cam])
        muL, PL = np.ascontiguousarray(self.lmk_mu[self.lmk]), np.ascontiguousarray(self.lmk_P[self.lmk])
        relin = np.zeros(self.F, np.int32)
        p = self.par
        c_d, c_i = ct.c_double, ct.c_int
        if local_relin:
            self.clk = (self.clk + 1) & 0xfffff
        self.lib.hm_sweep_factors(c_i(self.F), d(self.K), c_d(self.sigma2), c_d(p['nstds']), c_d(p['beta']), c_d(p['eta_damping']),
                                  c_i(p['num_undamped']), c_i(p['min_linear']), c_i(p['loss']), c_i(int(robustify)), c_i(int(local_relin)), c_i(0),
                                  c_i(self.clk), c_i(int(bool(local_relin))), d(self.x0), d(self.z), self.st.ctypes.data_as(_ip), d(self.avar), d(muC), d(PC), d(muL), d(PL),
                                  d(self.qC), d(self.qL), d(self.WC), d(self.VL), d(self.eC), d(self.MC), d(self.eL), d(self.ML),
                                  d(self.xt) if self.xt is not None else None, relin.ctypes.data_as(_ip))
        self.update_all_beliefs()
        return int(relin.sum())

    def beliefs(self):
        return self.cbel[:, :6], unpack(self.cbel[:, 6:], 6), self.lbel[:, :3], unpack(self.lbel[:, 3:], 3)

    def iters(self):
        return (self.clk - (self.st.view(np.uint32).astype(np.int64) >> 12)) & 0xfffff


def run_pair(hm, oracle_mod, name, sweeps, wf=50.0, float_impl=False, local_relin=True, **kw):
    from gbp_amd.balio import read_bal
    prob = read_bal(os.path.join(DATA, name), native=False)
    o = oracle_mod.OracleBA.from_problem(prob, threads=max(1, min(8, len(os.sched_getaffinity(0)))), **kw)
    h = HostBA(hm, prob, **kw)
    worst, n_relin = 0.0, 0
    weakening = np.log10(100.0) / 5
    for g in (o, h):
        g.generate_priors_var(wf)
        g.update_all_beliefs()
    for i in range(sweeps):
        for g in (o, h):
            if float_impl and (i + 1) % 2 == 0 and i < 10:
                g.weaken_priors(weakening)
            if i in (3, 8):
                g.set_iters_since_relin(1)
        o.synchronous_iteration(robustify=True, local_relin=local_relin)
        n_relin += h.synchronous_iteration(robustify=True, local_relin=local_relin)
        worst = max(worst, max(rel_err_rows(a, b) for a, b in zip(h.beliefs(), o.beliefs())))
        assert np.array_equal(h.iters(), o.relin_state()['iters_since_relin']), f"sweep {i + 1}: relinearisation bookkeeping differs"
    return worst, n_relin, h, o


@pytest.mark.parametrize('name', ['fr1desk.txt', 'fr2robot2.txt', 'fr1xyz_av.txt'])
def test_host_sweeps_match_the_oracle_through_two_relinearisations(hm, oracle_mod, name):
    worst, n_relin, h, o = run_pair(hm, oracle_mod, name, 30)
    assert n_relin > h.F                                       # sweeps 16 and 25 relinearise (nearly) every factor
    assert worst < 1e-6, worst


@pytest.mark.parametrize('loss', ['huber', 'constant'])
def test_host_sweeps_robust_losses(hm, oracle_mod, loss):
    worst, _, h, o = run_pair(hm, oracle_mod, 'fr1desk_vsmall.txt', 18, loss=loss, Nstds=3.0)
    st = o.relin_state()
    assert np.allclose(h.avar, st['adaptive_var'], rtol=1e-8) and np.array_equal((h.st >> 1) & 1, st['robust_flag'])
    assert worst < 1e-6, worst


def test_host_sweeps_prior_weakening(hm, oracle_mod):
    worst, _, _, _ = run_pair(hm, oracle_mod, 'fr1desk_vsmall.txt', 14, float_impl=True)
    assert worst < 1e-6, worst


def test_host_sweeps_damped_in_the_relinearising_sweep(hm, oracle_mod):
    """num_undamped_iters = 0: the dense remainder (Params::xtra) carries the out-of-span part of a damped eta."""
    worst, n_relin, _, _ = run_pair(hm, oracle_mod, 'fr1desk_vsmall.txt', 20, num_undamped_iters=0)
    assert n_relin > 0 and worst < 1e-6, (worst, n_relin)


def test_host_sweeps_global_damping(hm, oracle_mod):
    """local_relin=False: no relinearisation test, every message damped (gbp.py:52-54)."""
    worst, n_relin, _, _ = run_pair(hm, oracle_mod, 'fr1desk_vsmall.txt', 8, local_relin=False)
    assert n_relin == 0 and worst < 1e-6, worst


def test_relinearisation_clock_wraps(hm):
    """The state word stores the clock value of the last relinearisation (20 bits): ages survive the clock's wrap-around."""
    mask = (1 << 20) - 1
    for clk in (0, 3, mask - 1, mask):
        for iters in (0, 1, 7, (1 << 19) - 1):
            st = hm.hm_state_pack(iters, clk, 5, 1, 0, 0)
            assert hm.hm_state_age(st, clk) == iters
            assert hm.hm_state_age(st, (clk + 9) & mask) == iters + 9 or iters + 9 > mask
            assert (st >> 2) & 0x1ff == 5 and (st & 3) == 2
            st2 = hm.hm_state_set_age(st, 1, (clk + 2) & mask)
            assert hm.hm_state_age(st2, (clk + 2) & mask) == 1 and (st2 & 0xfff) == (st & 0xfff)
