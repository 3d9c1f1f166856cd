#!/bin/bash
# tools/ab_nt.sh -- A/B of nontemporal factor-stream loads / stores in the fused sweep (variants built into tools/ by hand:
#   hipcc ... -DGBP_NT_LOADS / -DGBP_NT_STORES -o tools/libgbp_<variant>.so), alternating runs, fresh process each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for rep in 1 2 3; do
  for v in "" NT_LOADS NT_STORES NT_BOTH; do
    lib=gbp_amd/libgbp_hip.so; [ -n "$v" ] && lib=tools/libgbp_$v.so
    GBP_HIP_LIB=$ROOT/$lib python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${v:-base}', round(d['value']), round(d['ms_per_step']*1e3,2), round(r['kernel_avg_ms']*1e3,2), round(r['reduce_avg_ms']*1e3,2))"
  done
done
