"""Drop-in `gbp.gbp_ba` / `vis` packages on the GPU: the exact call sequence of the reference's ba.py:68-105 (including
its Python loops over graph.factors and the viewer calls) must print the reference's ARE / energy trace (fixture G5)."""
import os
import sys

import numpy as np
import pytest

from conftest import DATA, REPO, belief_gap, golden

pytestmark = pytest.mark.gpu
COMPAT = os.path.join(REPO, 'gbp_amd', 'compat')


@pytest.fixture
def compat_path():
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in ('gbp', 'utils', 'vis')}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, COMPAT)
    yield
    sys.path.remove(COMPAT)
    for k in list(sys.modules):
        if k.split('.')[0] in ('gbp', 'utils', 'vis'):
            del sys.modules[k]
    sys.modules.update(saved)


def test_ba_script_sequence(compat_path):
    from gbp import gbp_ba
    import vis
    g5 = golden('G5_gate_small')
    configs = dict(gauss_noise_std=2, loss=None, Nstds=3.0, beta=0.01, num_undamped_iters=6, min_linear_iters=8,
                   eta_damping=0.4, prior_std_weaker_factor=50.0)
    graph = gbp_ba.create_ba_graph(os.path.join(DATA, 'fr1desk_small.txt'), configs)
    assert (len(graph.cam_nodes), len(graph.lmk_nodes), len(graph.factors)) == (20, 1216, 3917)
    graph.generate_priors_var(weaker_factor=50.0)
    graph.update_all_beliefs()
    scene = vis.ba_vis.create_scene(graph)
    viewer = vis.ba_vis.TrimeshSceneViewer(scene=scene, resolution=scene.camera.resolution)
    viewer.show()
    ares, energies, relins = [], [], []
    for i in range(30):
        if i == 3 or i == 8:
            for factor in graph.factors:
                factor.iters_since_relin = 1
        ares.append(graph.are())
        energies.append(graph.energy())
        relins.append(sum(1 for factor in graph.factors if factor.iters_since_relin == 0))
        viewer.update(graph)
        graph.synchronous_iteration(robustify=True, local_relin=True)
    assert relins == list(g5['n_relin'])
    assert np.allclose(ares, g5['are'], rtol=1e-6) and np.allclose(energies, g5['energy'], rtol=1e-5)
    bel = (np.array([n.belief.eta for n in graph.cam_nodes]), np.array([n.belief.lam for n in graph.cam_nodes]),
           np.array([n.belief.eta for n in graph.lmk_nodes]), np.array([n.belief.lam for n in graph.lmk_nodes]))
    assert belief_gap(bel, g5, 'it30_') < 1e-6
    # node / factor views
    f0 = graph.factors[0]
    assert f0.args[0].shape == (3, 3) and f0.adj_vIDs == [0, 20 + int(graph._lmk_of[0])]
    assert f0.messages[0].lam.shape == (6, 6) and f0.factor.lam.shape == (9, 9)
    assert np.allclose(graph.cam_nodes[0].Sigma @ graph.cam_nodes[0].belief.lam, np.eye(6), atol=1e-8)
    assert len(graph.cam_nodes[3].adj_factors) == int((graph._cam_of == 3).sum())
    assert viewer.n_updates == 30 and len(viewer.landmarks) == 1216
    assert abs(f0.reprojection_err() - np.linalg.norm(f0.compute_residual())) < 1e-12


def test_factor_graph_surface_of_the_device_graph(compat_path):
    """What scripts written against gbp.FactorGraph may touch beyond ba.py's own sequence: var_nodes is always
    a sequence, priors can be assigned through the node views, the batch joint is available, and the stage-wise methods
    (gbp.py:46-84) exist on the device graph too."""
    from gbp import gbp_ba
    configs = dict(gauss_noise_std=2, loss=None, Nstds=3.0, beta=0.01, num_undamped_iters=6, min_linear_iters=8,
                   eta_damping=0.4, prior_std_weaker_factor=50.0)
    graph = gbp_ba.create_ba_graph(os.path.join(DATA, 'fr1desk_vsmall.txt'), configs)
    assert len(graph.var_nodes) == 650 and graph.var_nodes[0] is graph.cam_nodes[0] and graph.var_nodes[10] is graph.lmk_nodes[0]
    assert sum(1 for _ in graph.var_nodes) == 650 and graph.var_nodes[-1] is graph.lmk_nodes[639]
    graph.generate_priors_var(weaker_factor=50.0)
    # writes through node.prior reach the device before the next call
    v = graph.lmk_nodes[5]
    lam0 = v.prior.lam.copy()
    v.prior.lam = 3.0 * lam0
    v.prior.eta = 3.0 * v.prior.eta
    assert np.allclose(graph.lmk_nodes[5].prior.lam, 3.0 * lam0)            # visible at once on the host side
    graph.update_all_beliefs()
    assert np.allclose(graph._engine.priors()[3][5], 3.0 * lam0)            # ... and on the device after the flush
    assert np.allclose(graph.lmk_nodes[5].belief.lam, 3.0 * lam0)           # no messages yet: belief = prior
    with pytest.raises(AttributeError):
        v.mu = np.zeros(3)
    # a prior written through the view and weakened right after (gbp_ba.py:36-42) is the weakened WRITTEN prior: the pending host
    # write reaches the device before weaken_priors / set_priors_var touch the device priors, and is not replayed over them later
    w = graph.lmk_nodes[9]
    lam9 = w.prior.lam.copy()
    w.prior.lam = 5.0 * lam9
    graph.weaken_priors(0.5)
    assert np.allclose(graph._engine.priors()[3][9], 2.5 * lam9)
    graph.update_all_beliefs()
    assert np.allclose(graph._engine.priors()[3][9], 2.5 * lam9) and np.allclose(graph.lmk_nodes[9].prior.lam, 2.5 * lam9)
    assert np.allclose(graph._engine.priors()[3][5], 1.5 * lam0)
    graph.weaken_priors(2.0)                                                  # back to where the checks below expect the priors
    graph.update_all_beliefs()
    # batch solution of the linearised problem (gbp.py:94-144) against the same thing assembled from the views
    eta, lam = graph.joint_distribution_inf()
    assert eta.shape == (60 + 1920,) and np.allclose(lam, lam.T)
    f7 = graph.factors[7]
    a, b = 6 * f7.adj_vIDs[0], 60 + 3 * (f7.adj_vIDs[1] - 10)
    blk = sum(graph.factors[int(k)].factor.lam[:6, :6] for k in np.nonzero(graph._cam_of == f7.adj_vIDs[0])[0])
    assert np.allclose(lam[a:a + 6, a:a + 6], blk + graph.cam_nodes[f7.adj_vIDs[0]].prior.lam, rtol=1e-10)
    mu, sigma = graph.joint_distribution_cov()
    assert np.isfinite(mu).all() and mu.shape == eta.shape
    # the four stages one by one = synchronous_iteration (gbp.py:86-92); the full stage-wise parity is tests/test_stagewise_gpu.py
    graph.robustify_all_factors()
    graph.relinearise_factors()
    graph.compute_all_messages(local_relin=True)
    graph.update_all_beliefs()
    graph.synchronous_iteration(robustify=True, local_relin=True)
    assert graph.count_relinearising() == sum(1 for f in graph.factors if f.iters_since_relin == 0)
    assert np.array_equal(graph.factors[3].linpoint, graph._engine.factors(3, 1, dense=False)['linpoint'][0])


def test_factor_views_held_across_device_calls(compat_path):
    """The reference's Factor objects are long-lived: `f = graph.factors[0]` (or `fs = list(graph.factors)`) taken once and used across
    synchronous_iteration() calls reads the CURRENT iters_since_relin, and a write through such a view reaches the sweep."""
    from gbp import gbp_ba
    configs = dict(gauss_noise_std=2, loss=None, Nstds=3.0, beta=0.01, num_undamped_iters=6, min_linear_iters=8,
                   eta_damping=0.4, prior_std_weaker_factor=50.0)
    graph = gbp_ba.create_ba_graph(os.path.join(DATA, 'fr1desk_vsmall.txt'), configs)
    graph.generate_priors_var(weaker_factor=50.0)
    graph.update_all_beliefs()
    f0 = graph.factors[0]
    held = list(graph.factors)
    assert f0.iters_since_relin == 1 and held[5].iters_since_relin == 1              # Factor.__init__, gbp.py:249
    graph.synchronous_iteration(robustify=True, local_relin=True)
    assert f0.iters_since_relin == 2 and held[5].iters_since_relin == 2              # read through views taken BEFORE the sweep
    assert np.array_equal(graph._engine.iters_since_relin(), [f.iters_since_relin for f in held])
    graph.synchronous_iteration(robustify=True, local_relin=True)
    held[7].iters_since_relin = 40                                                    # write through a stale-looking view ...
    f0.iters_since_relin = 12
    graph.synchronous_iteration(robustify=True, local_relin=True)
    dev = graph._engine.iters_since_relin()
    # ... is not lost: both factors were free to relinearise (>= min_linear_iters), so each is now at 0 (it did) or one past what was written
    assert dev[7] in (0, 41) and dev[0] in (0, 13) and dev[3] == 4
    assert held[7].iters_since_relin == dev[7] and f0.iters_since_relin == dev[0] and held[3].iters_since_relin == 4
    # the ba.py pattern still works after all that: a loop writes everywhere, the next loop reads what the sweep made of it
    for f in graph.factors:
        f.iters_since_relin = 1
    graph.synchronous_iteration(robustify=True, local_relin=True)
    assert all(f.iters_since_relin == 2 for f in held)
