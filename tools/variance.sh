#!/bin/bash
# tools/variance.sh <runs> [ENV=..]...  -- the default bench <runs> times in fresh processes: sweeps/s and steady kernel time of each
n=${1:-10}; shift
mkdir -p gpurun_out/var
out=""
for i in $(seq 1 $n); do
  env "$@" python bench.py --no-cpu-baseline > gpurun_out/var/$i.json 2> gpurun_out/var/$i.err || { out="$out FAIL"; continue; }
  out="$out $(python -c "
import json,sys; d=json.load(open('gpurun_out/var/$i.json')); print('%d/%.1f' % (round(d['value']), d['roofline']['kernel_avg_ms']*1e3))")"
done
echo "[$*] it/s / kernel us:$out"
