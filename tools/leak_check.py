#!/usr/bin/env python3
"""Create / run / destroy engines in a loop and watch the device's free memory (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gbp_amd.synthetic import make_synthetic
from gbp_amd.engine import BAEngine
from gbp_amd.linear import LinearEngine
import numpy as np
p = make_synthetic(n_cams=60, n_lmks=20000, obs_per_lmk=8, seed=1)
free0 = None
for i in range(60):
    for fused in (True, False):
        e = BAEngine.from_problem(p, fused=fused)
        e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(3); e.means_snapshot(); e.means_fetch(); blob = e.save_state(); e.load_state(blob)
        e.set_kernel_timing(1); e.iterate(2); e.sweep_clocks(); e.set_kernel_timing(0)
        e.peer_connect(0, [e.peer_export(1)]); e.update_beliefs_sharded(); e.iterate_sharded(2); e.sync()      # mailbox, merged exchange launch
        e.relinearise_factors(); e.compute_all_messages(); e.update_beliefs_sharded(); e.snapshot_state(); e.restore_snapshot()
        e.close()
    l = LinearEngine(np.arange(999), np.arange(1, 1000), np.zeros((999, 6)), np.tile(np.eye(6), (999, 1, 1)), np.zeros((1000, 3)), np.tile(np.eye(3), (1000, 1, 1)))
    l.update_all_beliefs(); l.iterate(2); l.close()
    free, total = torch.cuda.mem_get_info(0)
    if i == 4:
        free0 = free
    if i % 10 == 9:
        print(f"round {i + 1}: free {free / 2**20:.0f} MiB  (drift since round 5: {(free0 - free) / 2**20:+.1f} MiB)")
