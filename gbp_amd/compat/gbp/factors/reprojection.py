"""Pin-hole reprojection factor, host/numpy form (reference: gbp/factors/reprojection.py:12-44).

The GPU engine carries its own fp64 device version of these two functions (gbp_amd/csrc/gbp_math.hpp); this module
exists so that generic `gbp.Factor` objects and user code can still call them from Python."""
import numpy as np

from utils import transformations, lie_algebra, derivatives


def _camera_point(inp, K):
    R = lie_algebra.so3exp(inp[3:6])
    return R, K @ (R @ inp[6:9] + inp[0:3])


def meas_fn(inp, K):
    assert len(inp) == 9
    _, q = _camera_point(np.asarray(inp, dtype=float), K)
    return transformations.proj(q)


def jac_fn(inp, K):
    assert len(inp) == 9
    inp = np.asarray(inp, dtype=float)
    R, q = _camera_point(inp, K)
    JK = derivatives.proj_derivative(q) @ K
    return np.concatenate([JK, JK @ derivatives.dR_wx_dw(inp[3:6], inp[6:9]), JK @ R], axis=1)


if __name__ == '__main__':
    K = np.array([[517.306408, 0., 318.64304], [0., 516.469215, 255.313989], [0., 0., 1.]])
    derivatives.check_jac(jac_fn, np.random.rand(9), meas_fn, K)
