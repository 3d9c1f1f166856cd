#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/ (written to <out>/summary/)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, 'summary')
os.makedirs(dst, exist_ok=True)


def stats(name):
    for f in glob.glob(os.path.join(out, name, '**', '*kernel_stats.csv'), recursive=True):
        return f
    return None


def counter_means(name, kernel_sub):
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(out, name, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_sub in row['Kernel_Name']:
                a = agg[row['Counter_Name']]
                a[0] += 1
                a[1] += float(row['Counter_Value'])
    return {k: (s / n, n) for k, (n, s) in agg.items()}


def bench_line(path):
    try:
        for ln in open(path):
            if ln.startswith('{"metric"'):
                return json.loads(ln)
    except OSError:
        pass
    return None


def trace_timed_batch(name, sub, steps, warmup):
    """The K launches of kernel `sub` in bench.py's ONE timed batch (--single-batch: batch 0 is untimed, batch 1 is timed; each is
    W warm-up + K sweeps from the same restored state) -- the very sweeps bench.py's instrumented replay stamps, so the two averages
    must agree.  (Launches further on belong to the replays and their throw-away sweeps, which continue from wherever the state is.)"""
    import numpy as np
    for f in glob.glob(os.path.join(out, name, '**', '*kernel_trace.csv'), recursive=True):
        rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f)) if sub in r['Kernel_Name'])
        d = np.array([e - s for s, e in rows], dtype=float)[steps + 2 * warmup:2 * (steps + warmup)]
        if d.size == steps:
            med = float(np.median(d))
            return {"launches": int(d.size), "mean_us": float(d.mean()) / 1e3, "median_us": med / 1e3,
                    "steady_mean_us": float(d[d < 1.05 * med].mean()) / 1e3, "steady_launches": int((d < 1.05 * med).sum())}
    return None


def trace_summary(name, sub, per_batch):
    """From the kernel trace of a --stats run: durations of the launches of kernel `sub`, without the first batch (bench.py's untimed
    batch, clocks still ramping): mean, median, and the mean of the launches within 5 % of the median (steady sweeps: relinearising
    ones are 8-20 % longer)."""
    import numpy as np
    for f in glob.glob(os.path.join(out, name, '**', '*kernel_trace.csv'), recursive=True):
        rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f)) if sub in r['Kernel_Name'])
        d = np.array([e - s for s, e in rows], dtype=float)[per_batch:]
        if d.size:
            med = float(np.median(d))
            return {"launches": int(d.size), "mean_us": float(d.mean()) / 1e3, "median_us": med / 1e3,
                    "steady_mean_us": float(d[d < 1.05 * med].mean()) / 1e3, "steady_launches": int((d < 1.05 * med).sum())}
    return None


def kernel_avg_ns(stats_csv, sub):
    if not stats_csv:
        return None
    for row in csv.DictReader(open(stats_csv)):
        if sub in row['Name']:
            return float(row['AverageNs']), int(row['Calls'])
    return None


for size in ('1m', '10m', 'general'):
    f = stats('stats_' + size)
    if f:
        shutil.copy(f, os.path.join(dst, f'{tag}_kernel_stats_{size}.csv'))

traffic = {}
for size, F, L, C in (('1m', 1_000_000, 100_000, 500), ('10m', 10_000_000, 1_000_000, 500)):
    fe = counter_means(f'FETCH_SIZE_{size}', 'k_sweep_wat').get('FETCH_SIZE')
    wr = counter_means(f'WRITE_SIZE_{size}', 'k_sweep_wat').get('WRITE_SIZE')
    cal = counter_means(f'FETCH_SIZE_{size}', 'k_cam_reduce_tree').get('FETCH_SIZE')
    ka = kernel_avg_ns(stats('stats_' + size), 'k_sweep_wat')
    line = bench_line(os.path.join(out, f'bench_stats_{size}.log'))
    n_blocks = 256
    if fe and wr:
        fetch_b, write_b = fe[0] * 1000.0 * 2.0, wr[0] * 1000.0          # KB (x1000) -> bytes; gfx950: FETCH_SIZE reports half
        t = {"factors": F, "fetch_size_kb_raw": fe[0], "write_size_kb_raw": wr[0], "fetch_correction": 2.0,
             "traffic_bytes_per_launch": fetch_b + write_b, "read_bytes": fetch_b, "write_bytes": write_b,
             "dispatches": [fe[1], wr[1]]}
        if cal:
            # the reduce kernel reads exactly the workgroup tables: n_blocks x C rows of 28 doubles (27 sums + pad, gbp_fused.hpp TROW)
            t["calibration"] = {"kernel": "k_cam_reduce_tree", "expected_read_bytes": n_blocks * C * 28 * 8, "fetch_size_kb_raw": cal[0],
                                "ratio_expected_over_raw": n_blocks * C * 28 * 8 / (cal[0] * 1000.0)}
        if ka:
            t["kernel_avg_us_rocprofv3"] = ka[0] / 1e3
            t["kernel_calls"] = ka[1]
            ts = trace_summary('stats_' + size, 'k_sweep_wat', (line or {}).get('steps', 200) + (line or {}).get('warmup', 20))
            tb = trace_timed_batch('stats_' + size, 'k_sweep_wat', (line or {}).get('steps', 200), (line or {}).get('warmup', 20))
            if tb:
                t["rocprofv3_trace_timed_batch"] = tb
                t["layout_frac_on_rocprofv3_timed_batch_mean"] = line["roofline"]["bytes_per_launch"] / (tb["mean_us"] * 1e3) / 8000.0 if line else None
            if ts:
                t["rocprofv3_trace_without_first_batch"] = ts
                t["layout_frac_on_rocprofv3_steady_mean"] = line["roofline"]["bytes_per_launch"] / (ts["steady_mean_us"] * 1e3) / 8000.0 if line else None
                t["layout_frac_on_rocprofv3_mean_without_first_batch"] = line["roofline"]["bytes_per_launch"] / (ts["mean_us"] * 1e3) / 8000.0 if line else None
            t["hbm_gbs_on_traffic"] = (fetch_b + write_b) / ka[0]
            t["frac_of_8tbs"] = (fetch_b + write_b) / ka[0] / 8000.0
        if line:
            t["layout_bytes_per_launch"] = line["roofline"]["bytes_per_launch"]
            t["bench_kernel_avg_us_device_clock"] = line["roofline"]["kernel_avg_ms"] * 1e3
            t["bench_kernel_steady_us_device_clock"] = (line["roofline"].get("kernel_steady_ms") or 0.0) * 1e3
            t["bench_kernel_avg_us_hip_events_every_7th"] = (line["roofline"].get("kernel_event_ms") or 0.0) * 1e3
            t["bench_reduce_avg_us_device_clock"] = (line["roofline"].get("reduce_avg_ms") or 0.0) * 1e3
            t["bench_value_it_s"] = line["value"]
        traffic[size] = t
if traffic:
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gbp_amd', 'libgbp_hip.so')
    usage = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', f'{tag}_kernel_resource_usage.txt')
    doc = {"round": tag, "kernel": "k_sweep_wat<0, 8> (k_sweep_fused)",
           "library_sha256_16": hashlib.sha256(open(lib, 'rb').read()).hexdigest()[:16] if os.path.exists(lib) else None,
           "library_note": "the binary these passes measured; bench.py reports `traffic` only when it runs this very file",
           "kernel_resource_usage": [ln.strip() for ln in open(usage)] if os.path.exists(usage) else None,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh); FETCH_SIZE doubled "
                     "(gfx950, MI355X_MICROARCH.md 'HBM'), checked against k_cam_reduce_tree, whose read is exactly 256 x C x 28 doubles",
           "traffic_bytes_per_launch": traffic.get('1m', {}).get('traffic_bytes_per_launch'), "sizes": traffic}
    json.dump(doc, open(os.path.join(dst, f'{tag}_hbm_traffic.json'), 'w'), indent=1)

sq = {}
for d in sorted(glob.glob(os.path.join(out, 'sq_*'))):
    for k, (v, n) in counter_means(os.path.basename(d), 'k_sweep_wat').items():
        sq[k] = v
if sq:
    lines = [f"# SQ / TA counters of k_sweep_wat<0,8>, mean per dispatch, 1M-factor graph ({tag}; tools/profile_round.sh, one rocprofv3 --pmc pass per group)"]
    lines += [f"{k:34s} {v:.6g}" for k, v in sorted(sq.items())]
    if 'SQ_WAVE_CYCLES' in sq and sq.get('SQ_WAVE_CYCLES'):
        wc = sq['SQ_WAVE_CYCLES']
        for k in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM'):
            if k in sq:
                lines.append(f"{k + ' / SQ_WAVE_CYCLES':34s} {sq[k] / wc:.3f}")
    open(os.path.join(dst, f'{tag}_sq_counters.txt'), 'w').write('\n'.join(lines) + '\n')

abl = []
names = {0: 'full kernel', 1: 'no in-order wait before the camera accumulation', 4: 'no landmark-belief phase', 5: 'neither',
         8: 'camera records gathered from 8 cameras only (what a free gather would buy)', 12: 'cheap gather + no landmark-belief phase',
         14: 'cheap gather + no landmark-belief phase + no camera accumulation: streams, maths and stores only'}
for dbg in (0, 1, 4, 5, 8, 12, 14):
    ln = bench_line(os.path.join(out, f'bench_dbg{dbg}.json'))
    if ln:
        abl.append({"GBP_FUSED_DBG": dbg, "what": names[dbg], "kernel_avg_us": ln["roofline"]["kernel_avg_ms"] * 1e3,
                    "kernel_median_us": ln["roofline"]["kernel_median_ms"] * 1e3, "it_s": ln["value"]})
if abl:
    json.dump({"round": tag, "note": "timing only: the switched-off parts change the results (bit 2 = no accumulation changes convergence and with it the "
                                     "relinearisation work: compare medians)", "runs": abl}, open(os.path.join(dst, f'{tag}_ablations.json'), 'w'), indent=1)

ln = bench_line(os.path.join(out, 'bench_default.json'))
if ln:
    json.dump(ln, open(os.path.join(dst, f'{tag}_bench_default.json'), 'w'), indent=1)
for size in ('1m', '10m', 'general'):
    ln = bench_line(os.path.join(out, f'bench_stats_{size}.log'))
    if ln:
        json.dump(ln, open(os.path.join(dst, f'{tag}_bench_under_rocprof_{size}.json'), 'w'), indent=1)
print(open(os.path.join(dst, f'{tag}_hbm_traffic.json')).read() if traffic else 'no traffic data')
for f in sorted(os.listdir(dst)):
    print(' ', f)
