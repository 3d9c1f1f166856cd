#!/bin/bash
# tools/ab_libs.sh <reps> <lib-or-"default">...: the short bench line of several builds of the library, alternating, in one call
reps=$1; shift
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    if [ $v = default ]; then unset GBP_HIP_LIB; else export GBP_HIP_LIB=$PWD/tools/libgbp_$v.so; fi
    python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', f\"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {(r['kernel_steady_ms'] or 0)*1e3:.1f} reduce {(r.get('reduce_avg_ms') or 0)*1e3:.1f} parity {d['parity_check']['ok']} {d['parity_check'].get('camera_belief_gap_vs_reference')}\")"
  done
done
