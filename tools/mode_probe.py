#!/usr/bin/env python3
"""The 2M-factor sweep lands in one of two modes ~6 % apart from process to process (EXPERIMENTS.md rounds 5-6).  Is it WHERE the engine's
memory lies?  One process, several engines in a row, each behind a different amount of other device memory that stays allocated (so
that the engine's arena cannot come back to the pages the last one had): if the mode flips inside one process, it is placement.

    python tools/mode_probe.py [n_lmks]          (on the GPU box)
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from gbp_amd.engine import BAEngine
from gbp_amd.synthetic import make_synthetic

n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
p = make_synthetic(n_cams=500, n_lmks=n_l, obs_per_lmk=10, seed=0)
keep = []
for i, pad_mb in enumerate([0, 0, 700, 0, 1900, 300, 5000, 0]):
    if pad_mb:
        keep.append(torch.empty(pad_mb * 1024 * 1024, dtype=torch.uint8, device='cuda'))
    e = BAEngine.from_problem(p)
    e.generate_priors_var(50.0); e.update_all_beliefs(); e.sync(); e.snapshot_state()
    ts = []
    for rep in range(4):
        e.restore_snapshot(); e.iterate(5); e.sync()
        t0 = time.perf_counter(); e.iterate(20); e.sync(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
    rs = []
    for rep in range(10):                                     # a pure device-to-device copy of the same memory: does IT see the mode?
        e.sync(); t0 = time.perf_counter(); e.restore_snapshot(); e.sync(); rs.append((time.perf_counter() - t0) * 1e6)
    print(f"   restore_snapshot (streaming copy of the state): min {min(rs):7.1f} us median {np.median(rs):7.1f}", flush=True)
    print(f"engine {i}: {sum(len(k) for k in keep) >> 20:6d} MiB held in front of it, step {np.median(ts):7.1f} us (min {min(ts):7.1f})", flush=True)
    e.close()
