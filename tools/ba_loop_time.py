#!/usr/bin/env python3
"""Wall time of the reference's ba.py loop (ba.py:84-105: are(), energy(), the Python loop over graph.factors at i = 3, 8,
the viewer update, synchronous_iteration) through the drop-in packages on the GPU -- BASELINE config 2 as a user sees it."""
import os, sys, time
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(REPO, 'gbp_amd', 'compat')); sys.path.insert(0, REPO)
from gbp import gbp_ba
import vis
for name in ('fr1desk_small.txt', 'fr1desk.txt'):
    configs = dict(gauss_noise_std=2, loss=None, Nstds=3.0, beta=0.01, num_undamped_iters=6, min_linear_iters=8,
                   eta_damping=0.4, prior_std_weaker_factor=50.0)
    t0 = time.perf_counter()
    graph = gbp_ba.create_ba_graph(os.path.join(REPO, 'tests', 'golden', 'data', name), configs)
    graph.generate_priors_var(weaker_factor=50.0)
    graph.update_all_beliefs()
    t1 = time.perf_counter()
    scene = vis.ba_vis.create_scene(graph)
    viewer = vis.ba_vis.TrimeshSceneViewer(scene=scene, resolution=scene.camera.resolution)
    n = 200
    t2 = time.perf_counter()
    for i in range(n):
        if i == 3 or i == 8:
            for factor in graph.factors:
                factor.iters_since_relin = 1
        are, energy = graph.are(), graph.energy()
        viewer.update(graph)
        graph.synchronous_iteration(robustify=True, local_relin=True)
    are = graph.are()
    t3 = time.perf_counter()
    print(f"{name}: {len(graph.factors)} factors; set-up {1e3 * (t1 - t0):.1f} ms; ba.py loop {1e3 * (t3 - t2) / n:.3f} ms per iteration "
          f"(with are/energy/viewer), final ARE {are:.4f}")
