#!/usr/bin/env python3
"""Sweep time of a small shard (default 12 500 landmarks = 125k factors, the per-rank size at 8 GPUs) against the number of
workgroups of the fused sweep (GBP_FUSED_BLOCKS): fewer workgroups = fewer tables to write and reduce, more tiles per wave."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gbp_amd.synthetic import make_synthetic
from gbp_amd.engine import BAEngine
n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500
p = make_synthetic(n_cams=500, n_lmks=n_l, obs_per_lmk=10, seed=0)
for nb in (256, 192, 128, 96, 64, 32):
    os.environ['GBP_FUSED_BLOCKS'] = str(nb)
    e = BAEngine.from_problem(p)
    e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(50); e.sync()
    t0 = time.perf_counter(); e.iterate(400); e.sync(); dt = (time.perf_counter() - t0) / 400 * 1e6
    print(f"F={p.n_factors} workgroups {e.info()['n_blocks']:3d}: {dt:.1f} us/sweep", flush=True)
    e.close()
