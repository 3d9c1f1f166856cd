#!/bin/bash
# tools/size_sweep.sh <tag> [lmks...]: ns per factor of the fused sweep around the 256 MiB memory-side cache (bench.py --lmks N)
O=gpurun_out/${1:-size}; mkdir -p $O; shift
sizes=${@:-"70000 85000 100000 105000 110000 120000 135000 150000 200000 300000"}
for n in $sizes; do
  python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 --lmks $n 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']; F=d['config']['n_factors']
print(json.dumps(dict(nt=os.environ.get('GBP_FUSED_NT','auto'), n_factors=F, kernel_avg_us=round(r['kernel_avg_ms']*1e3,1), kernel_steady_us=round((r['kernel_steady_ms'] or 0)*1e3,1), ps_per_factor_steady=round((r['kernel_steady_ms'] or 0)*1e9/F,1), step_us=round(d['ms_per_step']*1e3,1), frac=round(r['frac'],3))))" | tee -a $O/size_sweep.jsonl
done
