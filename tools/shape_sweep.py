#!/usr/bin/env python3
"""Other shapes of a million factors (profiles/rNN_shape_sweep.json): observations per landmark 3 ... 100 at 500 cameras, one bench.py
line each (--steps 20 --warmup 5, no CPU baseline), printed as one JSON object.  Run on the GPU box.

    python tools/shape_sweep.py [obs ...]
"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obs_list = [int(a) for a in sys.argv[1:]] or [3, 5, 10, 20, 40, 100]
out = {}
for obs in obs_list:
    lmks = 1_000_000 // obs
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--lmks', str(lmks), '--obs', str(obs), '--steps', '20', '--warmup', '5',
                        '--no-cpu-baseline', '--no-hbm-size'], capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if not line:
        out[str(obs)] = {"error": r.stderr[-400:]}
        continue
    d = json.loads(line[0]); rf = d['roofline']
    out[str(obs)] = {"n_factors": d['config']['n_factors'], "n_lmks": lmks, "step_us": d['ms_per_step'] * 1e3, "sweep": d['config']['sweep'],
                     "kernel_avg_us": rf['kernel_avg_ms'] * 1e3, "kernel_steady_us": (rf.get('kernel_steady_ms') or 0) * 1e3,
                     "reduce_avg_us": (rf.get('reduce_avg_ms') or 0) * 1e3, "pack": os.environ.get('GBP_PACK', 'auto')}
print(json.dumps(out, indent=1))
