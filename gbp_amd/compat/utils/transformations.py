"""`proj` and `getT_axisangle`, the two functions of utils/transformations.py that a script or the viewer reaches
(reference lines 5-22); the quaternion / Euler helpers there are dead code."""
import numpy as np

from . import lie_algebra


def proj(x):
    x = np.asarray(x)
    if x.ndim == 1:
        return x[:-1] / x[-1]
    if x.ndim == 2:
        return x[:, :-1] / x[:, -1:]
    raise ValueError("proj expects a vector or a matrix of row vectors")


def getT_axisangle(x):
    T = np.eye(4)
    T[:3, :3] = lie_algebra.so3exp(x[3:6])
    T[:3, 3] = x[0:3]
    return T
