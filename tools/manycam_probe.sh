#!/bin/bash
# tools/manycam_probe.sh: shares of a graph with MANY random cameras (no locality): the automatic choice (camera sets where they fit)
# against the general sweep, step time and the reduce behind the sets
for cams in 1000 2000 5000; do for lm in 3000 6250 12500 25000; do
  for v in auto general; do
    extra=""; [ $v = general ] && extra="--no-fused"
    python bench.py --no-cpu-baseline --no-hbm-size --steps 100 --warmup 10 --cams $cams --lmks $lm $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$cams cams', d['config']['n_factors'], '$v', d['config']['sweep'], round(d['ms_per_step']*1e3,2), 'us  reduce', round((r.get('reduce_avg_ms') or 0)*1e3,2), r.get('reduce_kernel'), d['config'].get('camera_windows'))"
  done
done; done
