"""so(3) helpers the scripts reach through `utils.lie_algebra` (reference: utils/lie_algebra.py:11-42).

Only the two functions on the BA path are provided; se3exp / so3log / se3log of the reference are reached by no
script or config."""
import numpy as np

_EPS = np.finfo(float).eps


def S03_hat_operator(x):
    a, b, c = x
    return np.array([[0.0, -c, b], [c, 0.0, -a], [-b, a, 0.0]])


def so3exp(w):
    theta = float(np.sqrt(np.dot(w, w)))
    if theta < 3 * _EPS:                      # same cut-off as the reference (lie_algebra.py:37)
        return np.eye(3)
    W = S03_hat_operator(w)
    return np.eye(3) + (np.sin(theta) / theta) * W + ((1.0 - np.cos(theta)) / theta ** 2) * (W @ W)
