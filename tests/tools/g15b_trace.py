#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: where the engine, the C oracle and the reference part under `ba.py --float_implementation` at ba.py's default
length (fixture G15b = the reference's own run; VERDICT r5 item 6 asked for the sweep at which the gap peaks / crosses 1e-4).  Writes
gpurun_out/g15b_trace.json (copied to profiles/r06_g15b_gap_trace.json by hand): per file, the belief gap engine-reference and
oracle-reference at every checkpoint the reference reached, the reference's ARE there, the first sweep whose relinearisation count
differs, and the sweep the reference died in."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(HERE, '..'))
from conftest import DATA, belief_gap, golden      # noqa: E402
from gbp_amd.balio import read_bal                  # noqa: E402
from gbp_amd.engine import BAEngine                 # noqa: E402
from oracle import oracle                           # noqa: E402

out = {}
for tag in ('vsmall', 'small'):
    g = golden(f'G15b_floatimpl_200it_{tag}')
    p = read_bal(os.path.join(DATA, str(g['bal'])))
    cps = [int(c) for c in g['checkpoints'] if f'it{int(c)}_cam_eta' in g]
    n = min(len(g['are']), max(cps) + 1)
    rows = {}
    for name, make in (('engine_fused', lambda: BAEngine.from_problem(p, fused=True)), ('engine_general', lambda: BAEngine.from_problem(p, fused=False)),
                       ('oracle', lambda: oracle.OracleBA.from_problem(p))):
        e = make()
        e.generate_priors_var(50.0)
        e.update_all_beliefs()
        relin, gaps = [], {}

        def grab(i, graph):
            relin.append(graph.count_relinearising() if hasattr(graph, 'count_relinearising') else int((graph.relin_state()['iters_since_relin'] == 0).sum()))
            if i in cps:
                gaps[i] = belief_gap(graph.beliefs(), g, f'it{i}_')
        try:
            ares, _ = oracle.replay_ba(e, n, diagnostics=True, on_iter=grab, float_impl=True)
            err = None
        except Exception as ex:                                   # noqa: BLE001
            ares, err = np.zeros(0), f'{type(ex).__name__}: {ex}'
        m = min(len(relin), len(g['n_relin']))
        fork = np.nonzero(np.array(relin[:m]) != g['n_relin'][:m])[0]
        crossed = [k for k in sorted(gaps) if not gaps[k] < 1e-4]
        rows[name] = dict(belief_gap_vs_reference={str(k): float(v) for k, v in sorted(gaps.items())},
                          first_checkpoint_beyond_1e4=crossed[0] if crossed else None,
                          first_sweep_with_another_relinearisation_count=int(fork[0]) if fork.size else None, error=err)
        print(tag, name, {k: f'{v:.1e}' for k, v in sorted(gaps.items())}, 'fork', fork[:1], err)
        if hasattr(e, 'close'):
            e.close()
    out[tag] = dict(bal=str(g['bal']), reference_failed_in_sweep=int(g['reference_failed_in_sweep']), reference_error=str(g['reference_error']),
                    reference_are_at_checkpoints={str(k): float(g['are'][k]) for k in cps if k < len(g['are'])}, runs=rows)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/g15b_trace.json', 'w'), indent=1)
