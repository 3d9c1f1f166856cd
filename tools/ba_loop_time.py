#!/usr/bin/env python3
"""Wall time of the reference's ba.py loop body (ba.py:84-105, verbatim: the write loop over graph.factors at i = 3, 8, are(),
energy(), the read loop that counts relinearising factors, the viewer update, synchronous_iteration) through the drop-in packages on
the GPU -- what a user of the unchanged script sees per iteration, at the sizes of BASELINE configs 2, 3 and 4.

    python tools/ba_loop_time.py [--out gpurun_out/ba_loop.json]
"""
import argparse, json, os, sys, time
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(REPO, 'gbp_amd', 'compat')); sys.path.insert(0, REPO)
from gbp import gbp_ba
import vis
from gbp_amd.balio import read_bal
from gbp_amd.synthetic import make_synthetic

ap = argparse.ArgumentParser()
ap.add_argument('--out', default=None)
ap.add_argument('--iters', type=int, default=30)
args = ap.parse_args()
configs = dict(gauss_noise_std=2, loss=None, Nstds=3.0, beta=0.01, num_undamped_iters=6, min_linear_iters=8, eta_damping=0.4,
               prior_std_weaker_factor=50.0)
rows = []
for name in ('fr1desk_small.txt', 'fr1desk.txt', 'synthetic-1M'):
    t0 = time.perf_counter()
    if name.endswith('.txt'):
        graph = gbp_ba.create_ba_graph(os.path.join(REPO, 'tests', 'golden', 'data', name), configs)
    else:
        graph = gbp_ba.BAFactorGraph(make_synthetic(n_cams=500, n_lmks=100_000, obs_per_lmk=10, seed=0), configs)
    graph.generate_priors_var(weaker_factor=50.0)
    graph.update_all_beliefs()
    t1 = time.perf_counter()
    scene = vis.ba_vis.create_scene(graph)
    viewer = vis.ba_vis.TrimeshSceneViewer(scene=scene, resolution=scene.camera.resolution)
    t_write, t_read, t_diag, t_view, t_sweep, per_iter = [], [], [], [], [], []
    for i in range(args.iters):
        a = time.perf_counter()
        if i == 3 or i == 8:                                   # ba.py:91-93
            for factor in graph.factors:
                factor.iters_since_relin = 1
        b = time.perf_counter()
        are = graph.are()                                       # ba.py:95-96
        energy = graph.energy()
        c = time.perf_counter()
        n_factor_relins = 0                                     # ba.py:97-100
        for factor in graph.factors:
            if factor.iters_since_relin == 0:
                n_factor_relins += 1
        d = time.perf_counter()
        viewer.update(graph)                                    # ba.py:103
        e = time.perf_counter()
        graph.synchronous_iteration(robustify=True, local_relin=True)      # ba.py:105
        graph._engine.sync()
        f = time.perf_counter()
        if i in (3, 8):
            t_write.append(b - a)
        t_diag.append(c - b); t_read.append(d - c); t_view.append(e - d); t_sweep.append(f - e); per_iter.append(f - a)
    med = lambda x: sorted(x)[len(x) // 2] * 1e3 if x else 0.0
    row = dict(workload=name, n_factors=len(graph.factors), setup_ms=(t1 - t0) * 1e3, iteration_ms_median=med(per_iter), first_iteration_ms=per_iter[0] * 1e3,
               write_loop_ms=med(t_write), read_loop_ms=med(t_read), are_energy_ms=med(t_diag), viewer_ms=med(t_view), sweep_call_ms=med(t_sweep),
               last_relin_count=n_factor_relins, final_are=are)
    rows.append(row)
    print(json.dumps(row), flush=True)
if args.out:
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(rows, open(args.out, 'w'), indent=1)
