#!/usr/bin/env python3
"""Sequences (make_synthetic(window=...): every landmark seen from `obs` of `window` consecutive cameras, landmarks numbered along the
trajectory; closures: a share of them seen from anywhere along it) at one million factors: the fused sweep with per-workgroup camera
windows against the general sweep (--no-fused) and, where the whole table fits the LDS, against whole tables (GBP_WINDOWS=0).  One bench.py line each, printed as one JSON object
(profiles/rNN_camera_windows.json).  Run on the GPU box."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (cameras, window, observations per landmark[, landmarks[, share of the landmarks seen from anywhere along the trajectory]])
shapes = [(500, 30, 10), (2000, 30, 10), (2000, 100, 10), (10000, 30, 10), (13682, 60, 5, 616000), (2000, 30, 10, 100000, 0.02), (13682, 60, 5, 616000, 0.02)]
if len(sys.argv) > 1:
    shapes = [tuple(float(x) if '.' in x else int(x) for x in a.split(',')) for a in sys.argv[1:]]
out = {}
for sh in shapes:
    cams, window, obs = sh[:3]
    lmks = sh[3] if len(sh) > 3 else 1_000_000 // obs
    closures = sh[4] if len(sh) > 4 else 0.0
    for variant, extra, env in (("windows", [], {}), ("general", ['--no-fused'], {}), ("whole_tables", [], {"GBP_WINDOWS": "0"})):
        if variant == "whole_tables" and cams > 587:
            continue
        r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-hbm-size',
                            '--cams', str(cams), '--lmks', str(lmks), '--obs', str(obs), '--window', str(window), '--closures', str(closures)] + extra,
                           capture_output=True, text=True, env=dict(os.environ, **env))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        name = f"{cams}cams_window{window}_obs{obs}" + (f"_closures{closures}" if closures else "") + f"/{variant}"
        if not line:
            out[name] = {"error": r.stderr[-400:]}
            continue
        d = json.loads(line[0]); rf = d['roofline']
        out[name] = {"n_factors": d['config']['n_factors'], "step_us": d['ms_per_step'] * 1e3, "sweep": d['config']['sweep'],
                     "camera_windows": d['config'].get('camera_windows'), "kernel_avg_us": rf['kernel_avg_ms'] * 1e3,
                     "reduce_avg_us": (rf.get('reduce_avg_ms') or 0) * 1e3, "layout_bytes": rf['bytes_per_launch'], "frac": rf['frac'],
                     "parity": d['parity_check']}
        print(name, json.dumps(out[name]), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
