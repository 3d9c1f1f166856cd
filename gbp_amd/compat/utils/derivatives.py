"""Derivatives used by the reprojection factor (reference: utils/derivatives.py), numpy, host side."""
import numpy as np

from . import lie_algebra


def jac_fd(inp, meas_fn, *args, delta=1e-8):
    """Forward-difference Jacobian of meas_fn at inp (test helper)."""
    inp = np.asarray(inp, dtype=float)
    z0 = np.atleast_1d(meas_fn(inp, *args))
    J = np.empty((z0.shape[0], inp.shape[0]))
    for i in range(inp.shape[0]):
        step = inp.copy()
        step[i] += delta
        J[:, i] = (np.atleast_1d(meas_fn(step, *args)) - z0) / delta
    return J


def check_jac(jac_fn, inp, meas_fn, *args, threshold=1e-3):
    err = float(np.max(np.abs(jac_fn(inp, *args) - jac_fd(inp, meas_fn, *args))))
    if err < threshold:
        print(f"Passed! Jacobian correct to within {threshold}")
    else:
        print(f"Failed: maximum discrepancy to the finite-difference Jacobian {err} (threshold {threshold})")
    return err


def dR_wx_dw(w, x):
    """d(R(w) x)/dw = -R x^ (w w^T + (R^T - I) w^) / |w|^2   (Gallego & Yezzi, as in derivatives.py:36-45)."""
    R = lie_algebra.so3exp(w)
    inner = np.outer(w, w) + (R.T - np.eye(3)) @ lie_algebra.S03_hat_operator(w)
    return -(R @ lie_algebra.S03_hat_operator(x)) @ inner / np.dot(w, w)


def proj_derivative(x):
    x = np.asarray(x)
    if x.ndim == 1:
        n = x.shape[0] - 1
        return np.hstack([np.eye(n) / x[-1], (-x[:-1] / x[-1] ** 2)[:, None]])
