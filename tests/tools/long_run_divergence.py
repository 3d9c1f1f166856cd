#!/usr/bin/env python3
"""How far do three implementations of the same sweep drift apart over a LONG run?  (run on the GPU box)

fused HIP sweep, general HIP sweep and the CPU oracle start from identical states; the plain bench loop (no ba.py
schedule) runs `n` sweeps and the belief gaps are printed every `every` sweeps.  GBP with relinearisation thresholds is
chaotic: 1e-16 differences are amplified until a factor relinearises one sweep earlier in one implementation than in
the other, after which trajectories are different (equally valid) runs.  This tool shows WHEN that happens.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(HERE, '..'))

from conftest import DATA, rel_err_rows        # noqa: E402
from gbp_amd.balio import read_bal             # noqa: E402
from gbp_amd.engine import BAEngine            # noqa: E402
from oracle import oracle as om                # noqa: E402



def belief_gap(a, b):
    return max(rel_err_rows(x, y) for x, y in zip(a, b))


om.build()
name = sys.argv[1] if len(sys.argv) > 1 else 'fr1desk_small.txt'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
every = int(sys.argv[3]) if len(sys.argv) > 3 else 20
p = read_bal(os.path.join(DATA, name))
gs = {'fused': BAEngine.from_problem(p, fused=True), 'general': BAEngine.from_problem(p, fused=False),
      'oracle': om.OracleBA.from_problem(p)}
for g in gs.values():
    g.generate_priors_var(50.0)
    g.update_all_beliefs()
for i in range(0, n, every):
    for g in gs.values():
        for _ in range(every):
            g.synchronous_iteration(robustify=True, local_relin=True)
    b = {k: g.beliefs() for k, g in gs.items()}
    relin = {k: int((g.relin_state()['iters_since_relin'] == 0).sum()) for k, g in gs.items()}
    print(f"sweep {i + every:4d}  fused-oracle {belief_gap(b['fused'], b['oracle']):.2e}  general-oracle "
          f"{belief_gap(b['general'], b['oracle']):.2e}  fused-general {belief_gap(b['fused'], b['general']):.2e}  "
          f"ARE {gs['fused'].are():.4f} {gs['general'].are():.4f} {gs['oracle'].are():.4f}  relin {relin}")
