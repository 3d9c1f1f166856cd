#!/usr/bin/env python3
"""Wall time of the set-up path at the headline size (gbp_ba_create + priors + first beliefs), on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gbp_amd.synthetic import make_synthetic
from gbp_amd.engine import BAEngine
t0 = time.perf_counter(); p = make_synthetic(n_cams=500, n_lmks=100_000, obs_per_lmk=10, seed=0); t1 = time.perf_counter()
print(f"make_synthetic {t1 - t0:.2f} s")
for rep in range(3):
    t1 = time.perf_counter(); e = BAEngine.from_problem(p); e.sync(); t2 = time.perf_counter()
    e.generate_priors_var(50.0); e.sync(); t3 = time.perf_counter()
    e.update_all_beliefs(); e.sync(); t4 = time.perf_counter()
    print(f"create {t2 - t1:.3f} s  generate_priors {1e3 * (t3 - t2):.2f} ms  update_beliefs {1e3 * (t4 - t3):.2f} ms")
    e.close()
