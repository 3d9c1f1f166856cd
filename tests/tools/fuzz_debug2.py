#!/usr/bin/env python3
"""Where does one sweep of a fuzz seed lose digits?  Per-factor message gaps after sweep `k` (run on the GPU box)."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..')); sys.path.insert(0, os.path.join(HERE, '..'))
from test_fuzz_gpu import random_problem
from gbp_amd.engine import BAEngine
from gbp_amd.balio import reference_factor_order
from oracle import oracle as om
om.build()
seed, k = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
p = random_problem(seed)
loss = [None, 'huber', 'constant'][seed % 3]
cfg = dict(loss=loss, Nstds=float(rng.uniform(1.0, 3.0)), beta=float(rng.choice([0.005, 0.01, 0.05])),
           num_undamped_iters=int(rng.choice([1, 2, 6])), min_linear_iters=int(rng.choice([2, 4, 8])),
           eta_damping=float(rng.choice([0.3, 0.4, 0.7])), gauss_noise_std=float(rng.uniform(1.5, 3.0)))
flags = [(bool(rng.integers(0, 2)), bool(rng.random() < 0.8)) for _ in range(8)]
o = om.OracleBA.from_problem(p, threads=4, **cfg); e = BAEngine.from_problem(p, fused=False, **cfg)
for x in (o, e):
    x.generate_priors_var(30.0); x.update_all_beliefs()
order = reference_factor_order(p.cam_idx); cam, lmk = p.cam_idx[order], p.lmk_idx[order]
for i, (rob, rel) in enumerate(flags[:k + 1]):
    if i == k:
        so0 = o.relin_state(); bo0 = o.beliefs(); mo0 = o.messages(); cm0, lm0 = o.means(); ecm0, elm0 = e.means()
    for x in (o, e):
        x.synchronous_iteration(robustify=rob, local_relin=rel)
mo, me = o.messages(), e.messages()
so = o.relin_state()
gce = np.linalg.norm(mo[0] - me[0], axis=1) / np.maximum(np.linalg.norm(mo[0], axis=1), 1e-300)
gcl = np.linalg.norm((mo[1] - me[1]).reshape(len(cam), -1), axis=1) / np.linalg.norm(mo[1].reshape(len(cam), -1), axis=1)
gle = np.linalg.norm(mo[2] - me[2], axis=1) / np.maximum(np.linalg.norm(mo[2], axis=1), 1e-300)
worst = np.argsort(-np.maximum(np.maximum(gce, gcl), gle))[:8]
deg_l = np.bincount(lmk, minlength=p.n_lmks)
print("sweep", k, "flags", flags[k], "relinearised now:", int((so['iters_since_relin'] == 0).sum()), "of", len(cam))
for f in worst:
    cavL = bo0[3][lmk[f]] - mo0[3][f]; cavC = bo0[1][cam[f]] - mo0[1][f]
    from gbp_amd.synthetic import rodrigues
    pc = rodrigues(cm0[cam[f], 3:])[0] @ lm0[lmk[f]] + cm0[cam[f], :3]
    print(f"     point in camera frame {pc}  |w| {np.linalg.norm(cm0[cam[f], 3:]):.3f}  mean gaps cam {np.abs(cm0[cam[f]] - ecm0[cam[f]]).max():.1e} lmk {np.abs(lm0[lmk[f]] - elm0[lmk[f]]).max():.1e}  lmk mean {lm0[lmk[f]]}")
    print(f"  f={f} cam={cam[f]} lmk={lmk[f]} deg_l={deg_l[lmk[f]]} gaps eC {gce[f]:.1e} LC {gcl[f]:.1e} eL {gle[f]:.1e} iters {so['iters_since_relin'][f]} "
          f"avar {so['adaptive_var'][f]:.3g} robust {so['robust_flag'][f]} |eC| {np.linalg.norm(mo[0][f]):.2e} |eL| {np.linalg.norm(mo[2][f]):.2e} "
          f"cond cavL {np.linalg.cond(cavL):.1e} cavC {np.linalg.cond(cavC):.1e} eig cavL min {np.linalg.eigvalsh(cavL).min():.2e}")
