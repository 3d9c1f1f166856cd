"""ctypes wrapper around oracle/libgbp_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see gbp_oracle.c).  The product package gbp_amd/ never does.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libgbp_oracle.so')
_lib = None

LOSS = {None: 0, 'huber': 1, 'constant': 2}

_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int)
_bp = ct.POINTER(ct.c_ubyte)


# tests/test_sanitizers.py points this at a copy of the SAME source built with -fsanitize=address,undefined (a scratch directory)
_SAN_LIB = os.environ.get('GBP_ORACLE_SANITIZED_LIB')
if _SAN_LIB:
    _LIB_PATH = _SAN_LIB


def build(force=False):
    if _SAN_LIB:
        return _LIB_PATH                                      # built by the sanitizer test itself
    src = os.path.join(_HERE, 'gbp_oracle.c')
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libgbp_oracle.so'], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ct.CDLL(_LIB_PATH)
        L.gbpo_create.restype = ct.c_void_p
        L.gbpo_create.argtypes = [ct.c_int, ct.c_int, ct.c_int, _dp, _dp, _dp, _dp, _ip, _ip,
                                  ct.c_double, ct.c_int, ct.c_double, ct.c_double, ct.c_int, ct.c_int, ct.c_double]
        L.gbpo_destroy.argtypes = [ct.c_void_p]
        L.gbpo_set_threads.argtypes = [ct.c_void_p, ct.c_int]
        L.gbpo_generate_priors.argtypes = [ct.c_void_p, ct.c_double]
        L.gbpo_weaken_priors.argtypes = [ct.c_void_p, ct.c_double]
        L.gbpo_set_priors.argtypes = [ct.c_void_p, _dp, _dp]
        L.gbpo_update_beliefs.argtypes = [ct.c_void_p]
        L.gbpo_robustify.argtypes = [ct.c_void_p]
        L.gbpo_relinearise.argtypes = [ct.c_void_p]
        L.gbpo_compute_messages.argtypes = [ct.c_void_p, ct.c_int]
        L.gbpo_compute_factors.argtypes = [ct.c_void_p]
        L.gbpo_iterate.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int]
        L.gbpo_are.restype = ct.c_double
        L.gbpo_are.argtypes = [ct.c_void_p]
        L.gbpo_energy.restype = ct.c_double
        L.gbpo_energy.argtypes = [ct.c_void_p]
        L.gbpo_fn_eval.argtypes = [_dp, _dp, _dp, _dp]
        for name in ('gbpo_get_beliefs', 'gbpo_get_priors', 'gbpo_get_messages'):
            getattr(L, name).argtypes = [ct.c_void_p, _dp, _dp, _dp, _dp]
        L.gbpo_get_means.argtypes = [ct.c_void_p, _dp, _dp]
        L.gbpo_get_factors.argtypes = [ct.c_void_p, _dp, _dp, _dp, _ip, _ip, _dp]
        L.gbpo_get_relin_state.argtypes = [ct.c_void_p, _ip, _dp, _dp, _bp]
        L.gbpo_set_iters_since_relin.argtypes = [ct.c_void_p, _ip]
        L.gbpo_fill_iters_since_relin.argtypes = [ct.c_void_p, ct.c_int]
        L.gbpo_update_lmk_beliefs.argtypes = [ct.c_void_p]
        L.gbpo_cam_partial.argtypes = [ct.c_void_p, _dp]
        L.gbpo_cam_finish.argtypes = [ct.c_void_p, _dp, ct.c_int]
        L.gbpo_factor_lambda_max.argtypes = [ct.c_void_p, _dp, _dp]
        L.gbpo_set_prior_scalars.argtypes = [ct.c_void_p, _dp, _dp]
        L.gbpo_residual_sums.argtypes = [ct.c_void_p, _dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def fn_eval(x9, K4):
    """(meas_fn, jac_fn) of the reprojection factor at x9 with K4=(fx,fy,cx,cy)."""
    x = np.ascontiguousarray(x9, dtype=np.float64)
    K = np.ascontiguousarray(K4, dtype=np.float64)
    h = np.empty(2)
    J = np.empty((2, 9))
    lib().gbpo_fn_eval(_d(x), _d(K), _d(h), _d(J))
    return h, J


class OracleBA:
    """The reference's BAFactorGraph life cycle (ba.py:68-105) on the C restatement."""

    def __init__(self, K, cam_means, lmk_means, meas, cam_idx, lmk_idx, *, gauss_noise_std=2.0, loss=None,
                 Nstds=3.0, beta=0.01, num_undamped_iters=6, min_linear_iters=8, eta_damping=0.4, threads=1):
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(-1)
        if K.size == 9:
            K = np.array([K[0], K[4], K[2], K[5]])
        cam_means = np.ascontiguousarray(cam_means, dtype=np.float64)
        lmk_means = np.ascontiguousarray(lmk_means, dtype=np.float64)
        meas = np.ascontiguousarray(meas, dtype=np.float64)
        cam_idx = np.ascontiguousarray(cam_idx, dtype=np.int32)
        lmk_idx = np.ascontiguousarray(lmk_idx, dtype=np.int32)
        self.C, self.L, self.F = cam_means.shape[0], lmk_means.shape[0], meas.shape[0]
        self._h = lib().gbpo_create(self.C, self.L, self.F, _d(K), _d(cam_means), _d(lmk_means), _d(meas),
                                    _i(cam_idx), _i(lmk_idx), float(gauss_noise_std), LOSS[loss], float(Nstds),
                                    float(beta), int(num_undamped_iters), int(min_linear_iters), float(eta_damping))
        if not self._h:
            raise ValueError("gbpo_create failed (index out of range or out of memory)")
        lib().gbpo_set_threads(self._h, int(threads))

    @classmethod
    def from_problem(cls, p, **kw):
        return cls(p.K, p.cam_means, p.lmk_means, p.meas, p.cam_idx, p.lmk_idx, **kw)

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h and _lib is not None:
            _lib.gbpo_destroy(h)

    def set_threads(self, n):
        lib().gbpo_set_threads(self._h, int(n))

    def generate_priors_var(self, weaker_factor=100.0):
        lib().gbpo_generate_priors(self._h, float(weaker_factor))

    def weaken_priors(self, f):
        lib().gbpo_weaken_priors(self._h, float(f))

    def set_priors_var(self, cam_cov, lmk_cov):
        cc = np.ascontiguousarray(cam_cov, dtype=np.float64)
        lc = np.ascontiguousarray(lmk_cov, dtype=np.float64)
        lib().gbpo_set_priors(self._h, _d(cc), _d(lc))

    def update_all_beliefs(self):
        lib().gbpo_update_beliefs(self._h)

    def synchronous_iteration(self, local_relin=True, robustify=False):
        lib().gbpo_iterate(self._h, 1, int(robustify), int(local_relin))

    # the stages one by one (gbp.py:46-84)
    def robustify_all_factors(self):
        lib().gbpo_robustify(self._h)

    def relinearise_factors(self):
        lib().gbpo_relinearise(self._h)

    def compute_all_messages(self, local_relin=True):
        lib().gbpo_compute_messages(self._h, int(bool(local_relin)))

    def compute_all_factors(self):
        lib().gbpo_compute_factors(self._h)

    def iterate(self, n, robustify=True, local_relin=True):
        lib().gbpo_iterate(self._h, int(n), int(robustify), int(local_relin))

    def are(self):
        return float(lib().gbpo_are(self._h))

    def energy(self):
        return float(lib().gbpo_energy(self._h))

    def _four(self, fn, n_cam_like, n_lmk_like):
        ce, cl = np.empty((n_cam_like, 6)), np.empty((n_cam_like, 6, 6))
        le, ll = np.empty((n_lmk_like, 3)), np.empty((n_lmk_like, 3, 3))
        fn(self._h, _d(ce), _d(cl), _d(le), _d(ll))
        return ce, cl, le, ll

    def beliefs(self):
        return self._four(lib().gbpo_get_beliefs, self.C, self.L)

    def priors(self):
        return self._four(lib().gbpo_get_priors, self.C, self.L)

    def messages(self):
        return self._four(lib().gbpo_get_messages, self.F, self.F)

    def means(self):
        cm, lm = np.empty((self.C, 6)), np.empty((self.L, 3))
        lib().gbpo_get_means(self._h, _d(cm), _d(lm))
        return cm, lm

    def factors(self):
        eta, lam, lp = np.empty((self.F, 9)), np.empty((self.F, 9, 9)), np.empty((self.F, 9))
        cam, lmk, z = np.empty(self.F, np.int32), np.empty(self.F, np.int32), np.empty((self.F, 2))
        lib().gbpo_get_factors(self._h, _d(eta), _d(lam), _d(lp), _i(cam), _i(lmk), _d(z))
        return dict(eta=eta, lam=lam, linpoint=lp, cam=cam, lmk=lmk, z=z)

    def relin_state(self):
        it, d = np.empty(self.F, np.int32), np.empty(self.F)
        av, rb = np.empty(self.F), np.empty(self.F, np.uint8)
        lib().gbpo_get_relin_state(self._h, _i(it), _d(d), _d(av), rb.ctypes.data_as(_bp))
        return dict(iters_since_relin=it, eta_damping=d, adaptive_var=av, robust_flag=rb)

    def set_iters_since_relin(self, v):
        if np.isscalar(v):
            lib().gbpo_fill_iters_since_relin(self._h, int(v))
        else:
            a = np.ascontiguousarray(v, dtype=np.int32)
            assert a.shape == (self.F,)
            lib().gbpo_set_iters_since_relin(self._h, _i(a))


class OracleShard(OracleBA):
    """CPU stand-in for one rank's BAEngine in the landmark-sharded sweep (tests of gbp_amd.sharded only).

    Same five calls the sharded driver makes on the HIP engine (shard_begin / shard_end / factor_lambda_max /
    set_prior_scalars / residual_sums); the exchange buffer holds 42 doubles per camera (dense) instead of 27."""
    PARTIAL_DOUBLES = 42

    def factor_lambda_max(self):
        cm, lm = np.empty(self.C), np.empty(self.L)
        lib().gbpo_factor_lambda_max(self._h, _d(cm), _d(lm))
        return cm, lm

    def set_prior_scalars(self, cam_lambda, lmk_lambda):
        a = np.ascontiguousarray(cam_lambda, dtype=np.float64)
        b = np.ascontiguousarray(lmk_lambda, dtype=np.float64)
        lib().gbpo_set_prior_scalars(self._h, _d(a), _d(b))

    def shard_begin_host(self, with_messages=True, robustify=True, local_relin=True):
        if with_messages:
            if robustify:
                lib().gbpo_robustify(self._h)
            if local_relin:
                lib().gbpo_relinearise(self._h)
            lib().gbpo_compute_messages(self._h, int(local_relin))
        lib().gbpo_update_lmk_beliefs(self._h)
        out = np.empty(self.C * 42)
        lib().gbpo_cam_partial(self._h, _d(out))
        return out

    def shard_end_host(self, gathered, n_ranks):
        g = np.ascontiguousarray(gathered, dtype=np.float64)
        lib().gbpo_cam_finish(self._h, _d(g), int(n_ranks))

    def residual_sums(self):
        out = np.empty(2)
        lib().gbpo_residual_sums(self._h, _d(out))
        return out


def replay_ba(graph, n_iters, *, float_impl=False, final_prior_std_weaker_factor=100.0, num_weakening_steps=5,
              diagnostics=False, on_iter=None):
    """Drive any object with the BAFactorGraph surface through the loop of ba.py:84-105 (no viewer).

    `graph` needs weaken_priors / set_iters_since_relin / are / energy / synchronous_iteration.
    Returns (are[], energy[]) when diagnostics is on.  Shared by the oracle tests and the GPU tests
    so both sides follow exactly the same schedule (iters_since_relin reset at i=3 and i=8).
    """
    weakening = np.log10(final_prior_std_weaker_factor) / num_weakening_steps
    ares, energies = [], []
    for i in range(n_iters):
        if float_impl and (i + 1) % 2 == 0 and i < num_weakening_steps * 2:
            graph.weaken_priors(weakening)
        if i == 3 or i == 8:
            graph.set_iters_since_relin(1)
        if diagnostics:
            ares.append(graph.are())
            energies.append(graph.energy())
        if on_iter is not None:
            on_iter(i, graph)
        graph.synchronous_iteration(robustify=True, local_relin=True)
    return np.array(ares), np.array(energies)
