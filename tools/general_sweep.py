#!/usr/bin/env python3
"""The general sweep (more cameras than the fused sweep's LDS table holds, or --no-fused) at 1M factors and 500 / 1000 / 2000 / 3000
cameras, and a graph the size of the largest public BAL set (13 682 cameras, 3.08M factors): one bench.py line each, printed as one
JSON object (profiles/rNN_general_sweep.json).  Run on the GPU box."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = [("500_nofused", ['--cams', '500', '--no-fused']), ("1000", ['--cams', '1000']), ("2000", ['--cams', '2000']), ("3000", ['--cams', '3000']),
          ("13682_x_3.08M", ['--cams', '13682', '--lmks', '616000', '--obs', '5'])]
out = {}
for name, extra in shapes:
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-hbm-size'] + extra,
                       capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if not line:
        out[name] = {"error": r.stderr[-400:]}
        continue
    d = json.loads(line[0]); rf = d['roofline']
    out[name] = {"n_factors": d['config']['n_factors'], "n_cams": d['config']['n_cams'], "step_us": d['ms_per_step'] * 1e3, "sweep": d['config']['sweep'],
                 "factor_kernel_avg_us (HIP events, incl. ~3-5 us of dispatch)": rf['kernel_avg_ms'] * 1e3, "layout_bytes": rf['bytes_per_launch'],
                 "frac_of_8TBs_factor_kernel": rf['frac']}
print(json.dumps(out, indent=1))
