#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the fused sweep under the GBP_FUSED_DBG ablation switches (one PMC pass per counter and switch).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
# (the switches exist only in a library built with -DGBP_FUSED_DBG_SWITCHES: a scratch copy)
python -m gbp_amd.build --out "$ROOT/tools/libgbp_dbg.so" -DGBP_FUSED_DBG_SWITCHES > /dev/null
export GBP_HIP_LIB="$ROOT/tools/libgbp_dbg.so"
for dbg in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    GBP_FUSED_DBG=$dbg timeout 200 rocprofv3 --pmc $c -f csv -d gpurun_out/pmc_dbg${dbg}_$c -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    python - <<PY
import csv,glob
n=0;s=0.0
for f in glob.glob("gpurun_out/pmc_dbg${dbg}_$c/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sweep_wat" in r["Kernel_Name"]: n+=1; s+=float(r["Counter_Value"])
print("dbg=${dbg} $c sweep mean", s/max(n,1), "n", n)
PY
  done
done
