#!/usr/bin/env python3
"""How long is a kernel boundary of the sweep really?  A scratch build (-DGBP_END_STAMP) lets every workgroup of the sweep and of the reduce
kernel stamp the device clock when it is done (atomicMax: the last one's time stays), beside the start stamps of the product build:
    sweep start -> last sweep workgroup done -> reduce start -> last reduce workgroup done -> next sweep start.
The gaps are the end-of-kernel cache maintenance (dirty L2 lines go out: the XCDs' L2s are not coherent with each other) + the dispatch.

    python tools/boundary_probe.py [n_lmks ...]        (on the GPU box; builds tools/libgbp_endstamp.so if it is missing)
"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
LIB = os.path.join(HERE, 'libgbp_endstamp.so')
if not os.path.exists(LIB):
    from gbp_amd import build
    build.build(force=True, out=LIB, defines=['GBP_END_STAMP'])
os.environ['GBP_HIP_LIB'] = LIB
import numpy as np
from gbp_amd.engine import BAEngine
from gbp_amd.synthetic import make_synthetic
out = {}
for n_l in [int(a) for a in sys.argv[1:]] or [100_000, 12_500, 200_000]:
    p = make_synthetic(n_cams=500, n_lmks=n_l, obs_per_lmk=10, seed=0)
    e = BAEngine.from_problem(p)
    it, upd = e.iterate, e.update_all_beliefs
    if os.environ.get('PEER'):                               # the sharded loop with the one-rank peer exchange (tools/shard_probe.py `peer1`)
        e.peer_connect(0, [e.peer_export(1)])
        it, upd = e.iterate_sharded, e.update_beliefs_sharded
    e.generate_priors_var(50.0); upd(); e.sync(); e.snapshot_state()
    e.set_kernel_timing(1 << 30)
    for _ in range(20):
        e.restore_snapshot(); it(7)
    e.sync()
    c = e.sweep_clocks()
    c = c[-70:]
    nxt = np.append(c[1:, 0], np.nan)
    rows = dict(sweep_busy=c[:, 1] - c[:, 0], sweep_to_reduce=c[:, 2] - c[:, 1], reduce_busy=c[:, 3] - c[:, 2], reduce_to_next_sweep=nxt - c[:, 3])
    # (a restore sits between batches of seven: keep the gaps inside a batch)
    ok = np.isfinite(rows['reduce_to_next_sweep']) & (rows['reduce_to_next_sweep'] < 20)
    res = {k: float(np.nanmedian(v[ok])) for k, v in rows.items()}
    res['step'] = float(np.nanmedian((nxt - c[:, 0])[ok]))
    out[str(p.n_factors)] = res
    print(p.n_factors, {k: round(v, 2) for k, v in res.items()}, flush=True)
    e.close()
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/boundary_probe%s.json' % ('_peer' if os.environ.get('PEER') else ''), 'w'), indent=1)
