import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# The engine picks the staged (general) sweep on its own for graphs with few factors per camera (gbp_capi.hip: GBP_STAGED_BELOW); the
# tests name the sweep they mean (fused=True / False), so the automatic choice is off here and has a test of its own
# (tests/test_edge_shapes_gpu.py::test_sparse_graphs_take_the_staged_sweep).
os.environ.setdefault('GBP_STAGED_BELOW', '0')

GOLDEN = os.path.join(REPO, 'tests', 'golden')
DATA = os.path.join(GOLDEN, 'data')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def rel_err_rows(a, b):
    """max over the leading axis of ||a_i - b_i|| / ||b_i|| (Frobenius for matrices)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    a = a.reshape(a.shape[0], -1)
    b = b.reshape(b.shape[0], -1)
    num = np.linalg.norm(a - b, axis=1)
    den = np.linalg.norm(b, axis=1)
    den = np.where(den > 0, den, 1.0)
    return float((num / den).max()) if a.shape[0] else 0.0


def belief_gap(got, g, prefix):
    """SURVEY 8c parity metric over all variables: max relative error of eta and Lambda."""
    ce, cl, le, ll = got
    return max(rel_err_rows(ce, g[prefix + 'cam_eta']), rel_err_rows(cl, g[prefix + 'cam_lam']),
               rel_err_rows(le, g[prefix + 'lmk_eta']), rel_err_rows(ll, g[prefix + 'lmk_lam']))


@pytest.fixture(scope='session')
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle
