#!/bin/bash
# tools/sparse_probe.sh [lmks...]: small shares of the headline graph (500 random cameras, ten observations per landmark): step time of the
# automatic choice, whole tables (GBP_WINDOWS=0, fused forced), camera windows (GBP_WINDOWS=1, fused forced) and the general sweep
for lm in ${@:-1300 3000 6250 12500 25000 50000}; do
  for v in auto whole windows general; do
    unset GBP_WINDOWS GBP_STAGED_BELOW; extra=""
    case $v in whole) export GBP_WINDOWS=0 GBP_STAGED_BELOW=0;; windows) export GBP_WINDOWS=1 GBP_STAGED_BELOW=0;; general) extra="--no-fused";; esac
    python bench.py --no-cpu-baseline --no-hbm-size --steps 100 --warmup 10 --lmks $lm $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lm', '$v', d['config']['n_factors'], d['config']['sweep'], round(d['ms_per_step']*1e3,2), 'us', d['config'].get('camera_windows'))"
  done
done
