#!/usr/bin/env python3
"""Sweep time of the three loss variants of the fused kernel on the 1M-factor synthetic graph (secondary measurement)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gbp_amd.synthetic import make_synthetic
from gbp_amd.engine import BAEngine
p = make_synthetic(n_cams=500, n_lmks=100_000, obs_per_lmk=10, seed=0)
for loss in (None, 'huber', 'constant'):
    e = BAEngine.from_problem(p, loss=loss)
    e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(20); e.sync()
    t0 = time.perf_counter(); e.iterate(100); e.sync(); dt = time.perf_counter() - t0
    print(f"loss={loss}: {1e6 * dt / 100:.1f} us/sweep  ARE {e.are():.4f}")
    e.close()
