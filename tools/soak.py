#!/usr/bin/env python3
"""Soak: many sweeps on several graph shapes, watching for hangs (the fused sweep spins on LDS tickets) and NaNs; the last graph
runs past 2^20 sweeps: iters_since_relin saturates at 524 287 and the relinearisation clock of the state word wraps around."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gbp_amd.synthetic import make_synthetic
from gbp_amd.engine import BAEngine
for (c, l, o, n) in ((500, 100_000, 10, 40_000), (37, 5000, 7, 100_000), (516, 900, 30, 100_000), (700, 900, 30, 50_000), (8, 200, 5, 1_100_000)):
    p = make_synthetic(n_cams=c, n_lmks=l, obs_per_lmk=min(o, c), seed=c)
    e = BAEngine.from_problem(p)
    e.generate_priors_var(50.0); e.update_all_beliefs()
    t0 = time.perf_counter()
    for _ in range(n // 10_000):
        e.iterate(10_000); e.sync()
    dt = time.perf_counter() - t0
    ce, cl, le, ll = e.beliefs()
    ok = all(np.isfinite(a).all() for a in (ce, cl, le, ll))
    it = e.relin_state()['iters_since_relin']
    print(f"C={c} L={l} F={p.n_factors} cam_groups={e.info()['cam_groups']} iters_since_relin max {it.max()} min {it.min()}: {n} sweeps in {dt:.1f} s ({1e6 * dt / n:.1f} us each), finite={ok}, ARE {e.are():.4f}")
    e.close()
