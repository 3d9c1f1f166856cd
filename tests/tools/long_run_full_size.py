#!/usr/bin/env python3
"""The headline graph (500 cameras x 100k landmarks x 1M factors) through ba.py's schedule for N sweeps (default 200 = ba.py's own
default length), the engine against the C oracle: relinearisation count after every sweep, ARE, belief gap every tenth sweep.
Run on the GPU box (TEST INFRASTRUCTURE: about 135 s, most of it the oracle; the 26-sweep version is tests/test_hip_parity.py).

    python tests/tools/long_run_full_size.py [N] [out.json]
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(HERE, '..'))
from conftest import rel_err_rows                # noqa: E402
from gbp_amd.engine import BAEngine             # noqa: E402
from gbp_amd.synthetic import make_synthetic    # noqa: E402
from oracle import oracle as om                 # noqa: E402

om.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = make_synthetic(n_cams=500, n_lmks=100_000, obs_per_lmk=10, seed=0)
o = om.OracleBA.from_problem(p, threads=len(os.sched_getaffinity(0)))
e = BAEngine.from_problem(p)
for g in (o, e):
    g.generate_priors_var(50.0)
    g.update_all_beliefs()
rows, t0, first_fork = [], time.time(), None
for i in range(N):
    if i in (3, 8):                              # ba.py:91-93
        for g in (o, e):
            g.set_iters_since_relin(1)
    for g in (o, e):
        g.synchronous_iteration(robustify=True, local_relin=True)
    k, no, ne = i + 1, int((o.relin_state()['iters_since_relin'] == 0).sum()), e.count_relinearising()
    if no != ne and first_fork is None:
        first_fork = k
    if no != ne or k % 10 == 0:
        gap = max(rel_err_rows(a, b) for a, b in zip(e.beliefs(), o.beliefs()))
        ao, ae = o.are(), e.are()
        rows.append(dict(sweep=k, relin_oracle=no, relin_engine=ne, are_oracle=ao, are_engine=ae, belief_gap=gap))
        print(k, 'relin', no, ne, 'are', ao, ae, 'belief gap', f'{gap:.2e}', 't', round(time.time() - t0, 1), flush=True)
out = dict(what="1M-factor headline graph, ba.py schedule, fused HIP sweep against the C oracle (OpenMP)", sweeps=N, first_sweep_with_different_relinearisation_count=first_fork,
           max_belief_gap=max(r['belief_gap'] for r in rows), max_are_rel_gap=max(abs(r['are_engine'] - r['are_oracle']) / r['are_oracle'] for r in rows), rows=rows)
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'rows'}))
