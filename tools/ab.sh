#!/bin/bash
# tools/ab.sh "<env assignments A>" "<env assignments B>" ...  -- one short bench line per environment (kernel avg / median / min, sweeps/s).
# The same binary varies from process to process (about one run in five is ~20 % slower on the 1M-factor graph): repeat every
# setting several times before believing a difference.
mkdir -p gpurun_out/ab
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/ab/$i.json 2> gpurun_out/ab/$i.err || { echo "[$e] FAILED"; tail -3 gpurun_out/ab/$i.err; continue; }
  python - "$e" gpurun_out/ab/$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d['roofline']
print(f"[{sys.argv[1]}] {d['value']:.0f} it/s  step {d['ms_per_step']*1e3:.1f} us  kernel avg {r['kernel_avg_ms']*1e3:.1f} med {r['kernel_median_ms']*1e3:.1f} min {r['kernel_min_ms']*1e3:.1f} us  are {d['are_after']:.6f}")
PY
  grep "gbp ptrs" gpurun_out/ab/$i.err | head -1
done
