"""h(x1, x2) = x2 - x1 in any dimension (reference: gbp/factors/linear_displacement.py:8-14)."""
import numpy as np


def jac_fn(x):
    half = len(x) // 2
    eye = np.eye(half)
    return np.concatenate([-eye, eye], axis=1)


def meas_fn(x):
    x = np.asarray(x)
    half = len(x) // 2
    return x[half:2 * half] - x[:half]
