#!/bin/bash
# SQ occupancy / stall counters of the fused sweep (one pass per group; always under timeout: some counters hang rocprofv3).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp -f csv -d gpurun_out/pmc_sq_$i -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_sq_$i.log 2>&1
  echo "group $i ($grp) rc=$?"
  python - <<PY
import csv,glob
from collections import defaultdict
agg=defaultdict(lambda:[0,0.0])
for f in glob.glob("gpurun_out/pmc_sq_$i/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sweep_wat" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][0]+=1; agg[r["Counter_Name"]][1]+=float(r["Counter_Value"])
for k,(n,s) in sorted(agg.items()): print(f"  {k:28s} mean/dispatch {s/max(n,1):.6g}  (n={n})")
PY
done
