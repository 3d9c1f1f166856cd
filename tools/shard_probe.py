#!/usr/bin/env python3
"""One-GPU sweep time of the shards a 2 / 4 / 8-rank run gives every rank (500k / 250k / 125k factors of the headline graph's
family): the plain engine (gbp_ba_iterate: fused sweep + reduce-and-finish), the general sweep, and the in-library sharded
loop with the peer-store exchange at one rank (reduce -> mailbox -> finish that polls the arrival word: everything a rank
does per sweep except waiting for the others).  Writes gpurun_out/shard_probe.json (copied to profiles/ by hand).

    python tools/shard_probe.py [--sizes 50000 25000 12500] [--reps 400]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gbp_amd.synthetic import make_synthetic
from gbp_amd.engine import BAEngine

ap = argparse.ArgumentParser()
ap.add_argument('--sizes', type=int, nargs='*', default=[50_000, 25_000, 12_500])
ap.add_argument('--reps', type=int, default=400)
ap.add_argument('--out', default='gpurun_out/shard_probe.json')
ap.add_argument('--modes', nargs='*', default=['engine', 'engine_whole', 'general', 'peer1', 'peer1_whole'])      # _whole: whole camera tables (GBP_WINDOWS=0)
args = ap.parse_args()
rows = []
for n_l in args.sizes:
    p = make_synthetic(n_cams=500, n_lmks=n_l, obs_per_lmk=10, seed=0)
    for mode in args.modes:
        os.environ.pop('GBP_WINDOWS', None)
        if mode.endswith('_whole'):
            os.environ['GBP_WINDOWS'] = '0'
        if mode.endswith('_win'):                            # camera sets whenever they fit (GBP_WINDOWS=1), whatever the plan's threshold says
            os.environ['GBP_WINDOWS'] = '1'
        e = BAEngine.from_problem(p, fused=(mode not in ('general', 'peer1g')))
        if mode in ('peer1', 'peer1g', 'peer1_whole', 'peer1_win'):              # peer1g: the general sweep under the exchange (k_sweep_staged + k_cam_staged_xchg)
            e.peer_connect(0, [e.peer_export(1)])
            it, upd = e.iterate_sharded, e.update_beliefs_sharded
        else:
            it, upd = e.iterate, e.update_all_beliefs
        e.generate_priors_var(50.0); upd(); e.sync()
        e.snapshot_state()
        best = []
        for rep in range(3):
            e.restore_snapshot(); it(7); e.sync()            # steady sweeps only: nobody relinearises before sweep 9 ...
            ts = []
            for _ in range(args.reps // 8):                   # ... so time one sweep at a time, eight per restore
                e.restore_snapshot(); it(1); e.sync()
                t0 = time.perf_counter(); it(7); e.sync(); ts.append((time.perf_counter() - t0) / 7)
            best.append(sorted(ts)[len(ts) // 2])
        us = min(best) * 1e6
        # the same sweeps on the DEVICE's timeline: start stamp of every sweep kernel (workgroup 0 stores the constant-rate clock), so the
        # host's per-call costs -- the launch pipeline filling up and the sync at the end of every 7-sweep call, which in peer mode also
        # fetches the exchange's error word -- are not in it (round 6: they were 2-5 us of the host-timed figure above)
        import numpy as np
        e.set_kernel_timing(1 << 30)
        for _ in range(12):
            e.restore_snapshot(); it(1); it(7)
        e.sync()
        clk = e.sweep_clocks()
        e.set_kernel_timing(0)
        d = np.diff(clk[:, 0])
        d = d[(d > 0) & (d < 4 * us)]                        # (the restores between the calls are longer gaps)
        us_dev = float(np.median(d)) if d.size else float('nan')
        info = e.info()
        pi = e.plan_info()
        rows.append(dict(n_lmks=n_l, n_factors=p.n_factors, mode=mode, us_per_sweep=us, us_per_sweep_device_timeline=us_dev, n_blocks=info['n_blocks'], n_tiles=info['n_tiles'],
                         widest_camera_set=pi['max_window'], table_rows=pi['table_rows']))
        print(f"F={p.n_factors:8d} {mode:12s}: {us:7.1f} us/sweep host-timed, {us_dev:7.1f} on the device's timeline  (workgroups {info['n_blocks']}, tiles {info['n_tiles']}, table rows {pi['table_rows']}, widest set {pi['max_window']})", flush=True)
        e.close()
os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
json.dump(rows, open(args.out, 'w'), indent=1)
