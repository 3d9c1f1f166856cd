"""GPU parity of the linear pairwise GBP engine (include/gbp_lin.h, through the C ABI) against the numpy oracle and the
reference's ndim_posegraph.py trace (fixture G8).  fp64; the linear sweep has no thresholds, so differences stay at
rounding level: beliefs / messages asserted to 1e-9 relative."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

TOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def test_toy_posegraph_matches_reference_trace_and_oracle():
    from gbp_amd.linear import LinearEngine
    from oracle.linear_oracle import LinearOracle, toy_posegraph
    g8 = golden('G8_toy_linear')
    va, vb, fe, fl, fc, pe, pl = toy_posegraph(100, 3, 10, 1.0, seed=0)
    e = LinearEngine(va, vb, fe, fl, pe, pl, factor_const=fc)
    o = LinearOracle(va, vb, fe, fl, pe, pl, factor_const=fc)
    e.update_all_beliefs(); o.update_all_beliefs()
    assert rel(e.beliefs()[0], o.beliefs()[0]) < 1e-14
    mu_map = g8['n100d3_map_mu']
    energy, dist = [], []
    for _ in range(20):
        e.synchronous_iteration(); o.synchronous_iteration()
        energy.append(e.energy())
        dist.append(np.linalg.norm(e.get_means() - mu_map))
    assert np.allclose(energy, g8['n100d3_energy'], rtol=1e-6, atol=1e-3)
    assert np.allclose(dist, g8['n100d3_dist'], rtol=1e-5, atol=1e-5)
    assert np.allclose(e.get_means(), g8['n100d3_final_means'], rtol=1e-9, atol=1e-9)
    for a, b in zip(e.beliefs(), o.beliefs()):
        assert rel(a, b) < TOL
    for a, b in zip(e.messages(), o.messages()):
        assert rel(a, b) < TOL
    assert abs(e.energy() - o.energy()) < 1e-8 * abs(o.energy())


@pytest.mark.parametrize('D', [1, 2, 3, 4, 5, 6])
def test_random_pairwise_graphs_all_sizes(D):
    """Generic linear factors: random full-rank Jacobians over [a; b], random priors, damping 0.3, 15 sweeps in one call."""
    from gbp_amd.linear import LinearEngine
    from oracle.linear_oracle import LinearOracle
    rs = np.random.RandomState(10 + D)
    N, F = 40, 150
    va = rs.randint(0, N, F)
    vb = (va + 1 + rs.randint(0, N - 1, F)) % N
    fe, fl, fc = [], [], []
    for _ in range(F):
        J = rs.randn(D + 1, 2 * D)
        z = rs.randn(D + 1)
        fe.append(J.T @ z); fl.append(J.T @ J); fc.append(0.5 * z @ z)
    A = rs.randn(N, D, D)
    pl = A @ A.transpose(0, 2, 1) + 2.0 * np.eye(D)
    pe = rs.randn(N, D)
    e = LinearEngine(va, vb, np.array(fe), np.array(fl), pe, pl, factor_const=fc, eta_damping=0.3)
    o = LinearOracle(va, vb, np.array(fe), np.array(fl), pe, pl, factor_const=fc, eta_damping=0.3)
    e.update_all_beliefs(); o.update_all_beliefs()
    e.iterate(15); o.iterate(15)
    for a, b in zip(e.beliefs(), o.beliefs()):
        assert rel(a, b) < TOL
    for a, b in zip(e.messages(), o.messages()):
        assert rel(a, b) < TOL
    assert rel(e.get_means(), o.get_means()) < TOL
    assert abs(e.energy() - o.energy()) < 1e-8 * max(abs(o.energy()), 1.0)


def test_from_host_factor_graph_and_errors():
    """A graph built with the drop-in gbp.gbp classes (ndim_posegraph.py:67-91) moves to the device as is."""
    import os
    import sys
    from conftest import REPO
    from gbp_amd import _capi
    from gbp_amd.linear import LinearEngine
    compat = os.path.join(REPO, 'gbp_amd', 'compat')
    sys.path.insert(0, compat)
    try:
        from gbp import gbp
        from gbp.factors import linear_displacement
        rs = np.random.RandomState(3)
        n, dim = 30, 4
        mu0 = rs.rand(n, dim) * 10
        graph = gbp.FactorGraph(nonlinear_factors=False, eta_damping=0.2)
        for i in range(n):
            v = gbp.VariableNode(i, dim)
            v.prior.lam = np.eye(dim) / 3.0
            v.prior.eta = v.prior.lam @ mu0[i]
            graph.var_nodes.append(v)
        f = 0
        for i in range(n):
            for j in (i + 1, i + 5):
                if j < n:
                    a, b = graph.var_nodes[i], graph.var_nodes[j]
                    fac = gbp.Factor(f, [a, b], mu0[j] - mu0[i] + rs.normal(0, 0.3, dim), 0.3, linear_displacement.meas_fn,
                                     linear_displacement.jac_fn, loss=None, mahalanobis_threshold=2)
                    a.adj_factors.append(fac); b.adj_factors.append(fac); graph.factors.append(fac)
                    f += 1
        graph.update_all_beliefs()
        graph.compute_all_factors()
        e = graph.device_engine()
        assert isinstance(e, LinearEngine)
        e.update_all_beliefs()
        for _ in range(10):
            graph.synchronous_iteration()
        e.iterate(10)
        assert rel(e.get_means(), graph.get_means()) < TOL
        assert abs(e.energy() - graph.energy()) < 1e-8 * abs(graph.energy())
        eta, lam = e.beliefs()
        assert rel(eta, np.array([v.belief.eta for v in graph.var_nodes])) < TOL
        assert rel(lam, np.array([v.belief.lam for v in graph.var_nodes])) < TOL
    finally:
        sys.path.remove(compat)
        for m in [k for k in sys.modules if k == 'gbp' or k.startswith('gbp.') or k == 'utils' or k.startswith('utils.')]:
            del sys.modules[m]
    with pytest.raises(_capi.GbpError):
        LinearEngine([0], [0], np.zeros((1, 4)), np.zeros((1, 4, 4)), np.zeros((2, 2)), np.tile(np.eye(2), (2, 1, 1)))   # a == b
    with pytest.raises(_capi.GbpError):
        LinearEngine([0], [1], np.zeros((1, 14)), np.zeros((1, 14, 14)), np.zeros((2, 7)), np.tile(np.eye(7), (2, 1, 1)))  # dofs 7
    fresh = LinearEngine([0], [1], np.zeros((1, 4)), np.tile(np.eye(4), (1, 1, 1)), np.zeros((2, 2)), np.tile(np.eye(2), (2, 1, 1)))
    with pytest.raises(_capi.GbpError):
        fresh.iterate(1)                                                                                                   # no beliefs yet


def test_device_engine_prints_the_reference_stdout_trace():
    """The 'Iteration i // Energy ... // Av distance of means from MAP ...' lines of the reference's own
    `ndim_posegraph.py --n_varnodes 100 --dim 3 --n_iters 20` run (fixture G8 stdout), reproduced character for character
    with the sweeps on the device engine."""
    from gbp_amd.linear import LinearEngine
    from oracle.linear_oracle import toy_posegraph
    g8 = golden('G8_toy_linear')
    want = [ln for ln in str(g8['n100d3_stdout']).split('\n') if ln.startswith('Iteration')]
    va, vb, fe, fl, fc, pe, pl = toy_posegraph(100, 3, 10, 1.0, seed=0)
    e = LinearEngine(va, vb, fe, fl, pe, pl, factor_const=fc)
    e.update_all_beliefs()
    mu = g8['n100d3_map_mu']
    got = []
    for i in range(20):
        e.synchronous_iteration()
        got.append(f'Iteration {i}   //   Energy {e.energy():.4f}   //   '
                   f'Av distance of means from MAP {np.linalg.norm(e.get_means() - mu):4f}')     # ndim_posegraph.py:107-108
    assert got == want
