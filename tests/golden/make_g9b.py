#!/usr/bin/env python3
"""Fixture G9b (optional): the FULL headline graph (500 cams x 100k landmarks x 1M factors, the build's
generator with seed 0) through the reference itself for --sweeps sweeps (default 10) of the no-reset schedule bench.py
times: iters_since_relin starts at 1 (gbp.py:249), so nobody relinearises before sweep 8 (min_linear_iters = 8), sweep 8
relinearises every factor that moved by more than beta, sweeps 9 and 10 are the first undamped sweeps after it.  Stores the 500 camera beliefs and 2000 sampled landmark beliefs after
update_all_beliefs and after sweeps 1, 2, 8, 9, 10, the ARE after every sweep, the number of factors with
iters_since_relin == 0 after every sweep (ba.py:96-99) and, for 4000 sampled factors, iters_since_relin and eta_damping
after sweeps 9 and 10.

The reference's create_ba_graph scans all observations once per camera (gbp_ba.py:128-130: 5e8 Python iterations at this
size), so the graph is assembled here with the same classes in the same order (camera-major, file order inside a camera)
without that scan; everything else -- Factor.compute_factor, generate_priors_var, update_all_beliefs,
synchronous_iteration -- is the reference's own code.  Takes ~45 minutes and ~6 GB.  Run from the repo root:
    python tests/golden/make_g9b.py [--reference /root/reference]
"""
import argparse, os, sys, time
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument('--reference', default='/root/reference')
ap.add_argument('--lmks', type=int, default=100_000)
ap.add_argument('--sweeps', type=int, default=10)
ap.add_argument('--out', default=None)
args = ap.parse_args()
sys.dont_write_bytecode = True
REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, args.reference); sys.path.insert(0, REPO)
import warnings; warnings.simplefilter('ignore', SyntaxWarning)
from gbp import gbp_ba, gbp                      # the reference
from gbp.factors import reprojection
from gbp_amd.synthetic import make_synthetic

t0 = time.time()
p = make_synthetic(n_cams=500, n_lmks=args.lmks, obs_per_lmk=10, seed=0)
C, L, F = p.n_cams, p.n_lmks, p.n_factors
K = np.array([[p.K[0], 0, p.K[2]], [0, p.K[1], p.K[3]], [0, 0, 1.0]])
graph = gbp_ba.BAFactorGraph(eta_damping=0.4, beta=0.01, num_undamped_iters=6, min_linear_iters=8)
for i in range(C):                                # gbp_ba.py:114-119
    v = gbp_ba.FrameVariableNode(i, 6, i)
    v.mu = p.cam_means[i].copy()
    graph.cam_nodes.append(v)
for j in range(L):                                # gbp_ba.py:120-125
    v = gbp_ba.LandmarkVariableNode(C + j, 3, j)
    v.mu = p.lmk_means[j].copy()
    graph.lmk_nodes.append(v)
order = np.argsort(p.cam_idx, kind='stable')      # = the factor order of gbp_ba.py:128-130
for fid, o in enumerate(order):
    cam, lmk = graph.cam_nodes[p.cam_idx[o]], graph.lmk_nodes[p.lmk_idx[o]]
    f = gbp_ba.ReprojectionFactor(fid, [cam, lmk], p.meas[o], 2.0, None, 3.0, K)
    linpoint = np.concatenate((cam.mu, lmk.mu))
    f.compute_factor(linpoint)                    # gbp_ba.py:136-137
    cam.adj_factors.append(f); lmk.adj_factors.append(f)
    graph.factors.append(f)
    if fid % 100000 == 0:
        print(f"factor {fid} / {F}  ({time.time() - t0:.0f} s)", flush=True)
graph.n_factor_nodes, graph.n_edges = F, 2 * F
graph.var_nodes = graph.cam_nodes + graph.lmk_nodes
graph.n_var_nodes = C + L
print(f"graph built in {time.time() - t0:.0f} s", flush=True)
graph.generate_priors_var(weaker_factor=50.0)
graph.update_all_beliefs()
rs = np.random.RandomState(0)
sample = np.sort(rs.choice(L, size=min(2000, L), replace=False))
out = dict(lmk_sample=sample)

def snap(tag):
    out[tag + '_cam_eta'] = np.array([v.belief.eta for v in graph.cam_nodes])
    out[tag + '_cam_lam'] = np.array([v.belief.lam for v in graph.cam_nodes])
    out[tag + '_lmk_eta'] = np.array([graph.lmk_nodes[j].belief.eta for j in sample])
    out[tag + '_lmk_lam'] = np.array([graph.lmk_nodes[j].belief.lam for j in sample])
    out[tag + '_are'] = graph.are()

snap('it0')
fsample = np.sort(rs.choice(F, size=min(4000, F), replace=False))
out['factor_sample'] = fsample
are_trace, relin_trace = [out['it0_are']], []
for it in range(1, args.sweeps + 1):
    graph.synchronous_iteration(robustify=True, local_relin=True)
    are_trace.append(graph.are())
    relin_trace.append(sum(1 for f in graph.factors if f.iters_since_relin == 0))      # ba.py:96-99
    print(f"sweep {it} done ({time.time() - t0:.0f} s): ARE {are_trace[-1]:.6f}, relinearised {relin_trace[-1]}", flush=True)
    if it in (1, 2, 8, 9, 10):
        snap(f'it{it}')
    if it in (9, 10):
        out[f'it{it}_iters_since_relin'] = np.array([graph.factors[i].iters_since_relin for i in fsample], dtype=np.int32)
        out[f'it{it}_eta_damping'] = np.array([graph.factors[i].eta_damping for i in fsample])
out['are_trace'] = np.array(are_trace)
out['relin_trace'] = np.array(relin_trace, dtype=np.int64)
np.savez_compressed(args.out or os.path.join(REPO, 'tests', 'golden', f'G9b_synthetic_full_{F}.npz'), **out)
print("saved", {k: np.shape(v) for k, v in out.items()})
