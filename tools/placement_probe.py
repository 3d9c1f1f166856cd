#!/usr/bin/env python3
"""Is the ~20 % run-to-run bimodality of the fused sweep tied to the PROCESS or to the ALLOCATION?  One process builds the
1M-factor engine several times (fresh device allocations each time) and times 300 sweeps on each."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gbp_amd.synthetic import make_synthetic
from gbp_amd.engine import BAEngine
NL = int(os.environ.get('LMKS', 100_000))
p = make_synthetic(n_cams=500, n_lmks=NL, obs_per_lmk=10, seed=0)
keep = []
shared = None
if os.environ.get('SHARED_STREAM'):
    import torch
    shared = torch.cuda.Stream()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    e = BAEngine.from_problem(p)
    if shared is not None:
        e.set_stream(shared.cuda_stream)
    e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(60); e.sync()
    ts = []
    for k in range(3):
        t0 = time.perf_counter(); e.iterate(200); e.sync(); ts.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"engine {rep}: {ts[0]:.1f} {ts[1]:.1f} {ts[2]:.1f} us/sweep = {min(ts) / (NL / 1e5):.1f} us per 1M factors", flush=True)
    if rep % 2 == 0:
        keep.append(e)            # hold on to some engines so that the next one lands elsewhere
    else:
        e.close()
