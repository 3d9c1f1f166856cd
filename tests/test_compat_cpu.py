"""Drop-in packages, host side: ndim_posegraph.py (BASELINE config 1, CPU plumbing) runs UNCHANGED from the reference
checkout against gbp_amd/compat and reproduces the reference's own stdout trace (fixture G8)."""
import io
import os
import contextlib
import runpy
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, golden

REF = '/root/reference'
COMPAT = os.path.join(REPO, 'gbp_amd', 'compat')


def _restore(mods):
    for k in list(sys.modules):
        if k.split('.')[0] in ('gbp', 'utils', 'vis'):
            del sys.modules[k]
    sys.modules.update(mods)


@pytest.fixture
def compat_path():
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in ('gbp', 'utils', 'vis')}
    _restore({})
    sys.path.insert(0, COMPAT)
    yield
    sys.path.remove(COMPAT)
    _restore(saved)


def test_generic_factor_graph_matches_reference_trace(compat_path):
    """Re-creates ndim_posegraph.py's graph construction (its published CLI: --n_varnodes 100 --dim 3 --n_iters 20) with
    the drop-in classes; energies and distances to the batch MAP must match fixture G8 (reference stdout)."""
    from gbp import gbp
    from gbp.factors import linear_displacement
    g8 = golden('G8_toy_linear')
    np.random.seed(0)
    n, dim, M, std, iters = 100, 3, 10, 1.0, 20
    priors_mu = np.random.rand(n, dim) * 10
    prior_lambda = np.linalg.inv(3 * np.eye(dim))
    pairs, meas = [], []
    for i, mu in enumerate(priors_mu):
        d = np.array([np.linalg.norm(mu - m1) for m1 in priors_mu])
        for j in d.argsort()[1:M + 1]:
            if [j, i] not in pairs:
                meas.append(mu - priors_mu[j] + np.random.normal(0., std, dim))
                pairs.append([i, j])
    graph = gbp.FactorGraph(nonlinear_factors=False)
    for i in range(n):
        v = gbp.VariableNode(i, dim)
        v.prior.eta, v.prior.lam = prior_lambda @ priors_mu[i], prior_lambda
        graph.var_nodes.append(v)
    for f, z in enumerate(meas):
        a, b = graph.var_nodes[pairs[f][0]], graph.var_nodes[pairs[f][1]]
        fac = gbp.Factor(f, [a, b], z, std, linear_displacement.meas_fn, linear_displacement.jac_fn, loss=None, mahalanobis_threshold=2)
        a.adj_factors.append(fac); b.adj_factors.append(fac)
        graph.factors.append(fac)
    graph.update_all_beliefs()
    graph.compute_all_factors()
    mu, _ = graph.joint_distribution_cov()
    assert np.allclose(mu, g8['n100d3_map_mu'], rtol=1e-9, atol=1e-9)
    energy, dist = [], []
    for _ in range(iters):
        graph.synchronous_iteration()
        energy.append(graph.energy())
        dist.append(np.linalg.norm(graph.get_means() - mu))
    assert np.allclose(energy, g8['n100d3_energy'], rtol=1e-6, atol=1e-3)      # fixture values are the printed 4 decimals
    assert np.allclose(dist, g8['n100d3_dist'], rtol=1e-5, atol=1e-5)
    assert np.allclose(graph.get_means(), g8['n100d3_final_means'], rtol=1e-8, atol=1e-8)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'ndim_posegraph.py')), reason="reference checkout not mounted")
def test_reference_script_runs_unchanged_on_dropin_packages():
    env = dict(os.environ, PYTHONPATH=COMPAT + os.pathsep + REPO, PYTHONDONTWRITEBYTECODE='1')
    out = subprocess.run([sys.executable, os.path.join(REF, 'ndim_posegraph.py')], env=env, cwd=REPO, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    g8 = golden('G8_toy_linear')
    lines = [ln for ln in out.stdout.split('\n') if ln.startswith('Iteration')]
    energy = np.array([float(ln.split('Energy')[1].split('//')[0]) for ln in lines])
    dist = np.array([float(ln.split('MAP')[1]) for ln in lines])
    assert len(lines) == 50
    assert np.allclose(energy, g8['defaults_energy'], atol=2e-4) and np.allclose(dist, g8['defaults_dist'], atol=2e-6)


def test_reprojection_module_against_g1(compat_path):
    from gbp.factors import reprojection
    g = golden('G1_reproj_fn')
    for x, h, J in zip(g['x'][:64], g['h'], g['J']):
        assert np.allclose(reprojection.meas_fn(x, g['K']), h, rtol=1e-12, atol=1e-10)
        assert np.allclose(reprojection.jac_fn(x, g['K']), J, rtol=1e-10, atol=1e-9 * np.abs(J).max())


def test_read_balfile_signature(compat_path):
    from utils import read_balfile
    out = read_balfile.read_balfile(os.path.join(REPO, 'tests', 'golden', 'data', 'fr1desk_vsmall.txt'))
    assert out[:3] == (10, 640, 1801) and out[8].shape == (3, 3) and isinstance(out[6], list)
