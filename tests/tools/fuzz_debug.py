#!/usr/bin/env python3
"""Per-sweep gap trace of one fuzz seed (tests/test_fuzz_gpu.py) -- run on the GPU box."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..')); sys.path.insert(0, os.path.join(HERE, '..'))
from conftest import rel_err_rows
from test_fuzz_gpu import random_problem
from gbp_amd.engine import BAEngine
from oracle import oracle as om
om.build()
for seed in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    p = random_problem(seed)
    loss = [None, 'huber', 'constant'][seed % 3]
    cfg = dict(loss=loss, Nstds=float(rng.uniform(1.0, 3.0)), beta=float(rng.choice([0.005, 0.01, 0.05])),
               num_undamped_iters=int(rng.choice([1, 2, 6])), min_linear_iters=int(rng.choice([2, 4, 8])),
               eta_damping=float(rng.choice([0.3, 0.4, 0.7])), gauss_noise_std=float(rng.uniform(1.5, 3.0)))
    flags = [(bool(rng.integers(0, 2)), bool(rng.random() < 0.8)) for _ in range(8)]
    print(f"seed {seed}: C={p.n_cams} L={p.n_lmks} F={p.n_factors} cfg={cfg}")
    o = om.OracleBA.from_problem(p, threads=4, **cfg); e = BAEngine.from_problem(p, fused=True, **cfg); g = BAEngine.from_problem(p, fused=False, **cfg)
    for x in (o, e, g):
        x.generate_priors_var(30.0); x.update_all_beliefs()
    print("  init gap", max(rel_err_rows(a, b) for a, b in zip(e.beliefs(), o.beliefs())))
    for i, (rob, rel) in enumerate(flags):
        for x in (o, e, g):
            x.synchronous_iteration(robustify=rob, local_relin=rel)
        so, se = o.relin_state(), e.relin_state()
        gaps = [rel_err_rows(a, b) for a, b in zip(e.beliefs(), o.beliefs())]
        gg = max(rel_err_rows(a, b) for a, b in zip(g.beliefs(), o.beliefs()))
        ce, cl, le, ll = o.beliefs()
        mineig_c = min(np.linalg.eigvalsh(m).min() for m in cl); mineig_l = min(np.linalg.eigvalsh(m).min() for m in ll)
        print(f"  sweep {i} rob={rob} relin={rel}: fused gaps {['%.1e' % v for v in gaps]} general {gg:.1e}  iters mismatch "
              f"{int((so['iters_since_relin'] != se['iters_since_relin']).sum())} robust mismatch {int((so['robust_flag'] != se['robust_flag']).sum())}"
              f"  min eig cam {mineig_c:.2e} lmk {mineig_l:.2e}  ARE {o.are():.3g}")
