"""Linear pairwise GBP on the MI355X (include/gbp_lin.h): the reference's generic FactorGraph path
(gbp/gbp.py with nonlinear_factors=False, ndim_posegraph.py) for two-variable factors over d <= 6 dofs.

`LinearEngine.from_factor_graph(graph)` takes a host graph built with the drop-in `gbp.gbp` classes
(after `compute_all_factors()`, ndim_posegraph.py:91) and runs its sweeps on the device; the host
graph is left untouched.  No CPU fallback: without a GPU `gbp_lin_create` fails with GBP_ENODEV.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _capi
from ._capi import check, dptr, iptr, f64, i32


class LinearEngine:
    def __init__(self, var_a, var_b, factor_eta, factor_lam, prior_eta, prior_lam, factor_const=None, eta_damping=0.0, device=0):
        self._lib = _capi.load()
        prior_eta = f64(prior_eta)
        self.N, self.D = prior_eta.shape
        prior_lam = f64(prior_lam, (self.N, self.D, self.D))
        var_a, var_b = i32(var_a).reshape(-1), i32(var_b).reshape(-1)
        self.F = var_a.shape[0]
        factor_eta = f64(factor_eta, (self.F, 2 * self.D))
        factor_lam = f64(factor_lam, (self.F, 2 * self.D, 2 * self.D))
        fc = None if factor_const is None else f64(factor_const, (self.F,))
        if var_b.shape[0] != self.F:
            raise ValueError("var_a / var_b length mismatch")
        d = _capi.LinDesc()
        d.n_vars, d.dofs, d.n_factors, d.device = self.N, self.D, self.F, int(device)
        d.var_a, d.var_b = iptr(var_a), iptr(var_b)
        d.factor_eta, d.factor_lam, d.factor_const = dptr(factor_eta), dptr(factor_lam), dptr(fc)
        d.prior_eta, d.prior_lam = dptr(prior_eta), dptr(prior_lam)
        d.eta_damping = float(eta_damping)
        self._h = ct.c_void_p()
        check(self._lib.gbp_lin_create(ct.byref(self._h), ct.byref(d)))

    @classmethod
    def from_factor_graph(cls, graph, device=0):
        """A host FactorGraph(nonlinear_factors=False) of pairwise equal-size factors (gbp/gbp.py:11-153) -> device."""
        if getattr(graph, 'nonlinear_factors', True):
            raise ValueError("only linear graphs (nonlinear_factors=False) have a device path here; BA graphs use gbp.gbp_ba")
        index = {v.variableID: i for i, v in enumerate(graph.var_nodes)}
        dofs = {v.dofs for v in graph.var_nodes}
        if len(dofs) != 1:
            raise ValueError("all variables must have the same number of dofs")
        D = dofs.pop()
        va, vb, fe, fl, fc = [], [], [], [], []
        for fac in graph.factors:
            if len(fac.adj_vIDs) != 2:
                raise ValueError("only two-variable factors")
            va.append(index[fac.adj_vIDs[0]]); vb.append(index[fac.adj_vIDs[1]])
            fe.append(np.asarray(fac.factor.eta, dtype=float)); fl.append(np.asarray(fac.factor.lam, dtype=float))
            x0 = np.asarray(fac.linpoint, dtype=float)
            J = np.atleast_2d(fac.jac_fn(fac.linpoint, *fac.args))
            r = J @ x0 + fac.measurement - fac.meas_fn(fac.linpoint, *fac.args)        # = z - h(0) for a linear h
            fc.append(0.5 * float(np.dot(np.atleast_1d(r), np.atleast_1d(r))) / fac.adaptive_gauss_noise_var)
        for v in graph.var_nodes:                                                       # adjacency order the device assumes
            ids = [f.factorID for f in v.adj_factors]
            if ids != sorted(ids):
                raise ValueError("adj_factors must be in ascending factor id (ndim_posegraph.py:86-88)")
        return cls(va, vb, np.array(fe).reshape(-1, 2 * D), np.array(fl).reshape(-1, 2 * D, 2 * D),
                   np.array([v.prior.eta for v in graph.var_nodes]).reshape(-1, D),
                   np.array([v.prior.lam for v in graph.var_nodes]).reshape(-1, D, D),
                   factor_const=fc, eta_damping=graph.eta_damping, device=device)

    def close(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._lib.gbp_lin_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # FactorGraph surface (gbp.py:36-58, 86-92, 146-153)
    def update_all_beliefs(self):
        check(self._lib.gbp_lin_update_beliefs(self._h))

    def synchronous_iteration(self):
        check(self._lib.gbp_lin_iterate(self._h, 1))

    def iterate(self, n):
        check(self._lib.gbp_lin_iterate(self._h, int(n)))

    def energy(self):
        out = ct.c_double()
        check(self._lib.gbp_lin_energy(self._h, ct.byref(out)))
        return out.value

    def get_means(self):
        mu = np.empty((self.N, self.D))
        check(self._lib.gbp_lin_get_means(self._h, dptr(mu)))
        return mu.reshape(-1)

    def beliefs(self):
        eta, lam = np.empty((self.N, self.D)), np.empty((self.N, self.D, self.D))
        check(self._lib.gbp_lin_get_beliefs(self._h, dptr(eta), dptr(lam)))
        return eta, lam

    def messages(self):
        ea, la = np.empty((self.F, self.D)), np.empty((self.F, self.D, self.D))
        eb, lb = np.empty((self.F, self.D)), np.empty((self.F, self.D, self.D))
        check(self._lib.gbp_lin_get_messages(self._h, dptr(ea), dptr(la), dptr(eb), dptr(lb)))
        return ea, la, eb, lb

    def sync(self):
        check(self._lib.gbp_lin_sync(self._h))
