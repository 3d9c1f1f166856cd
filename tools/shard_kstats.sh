#!/bin/bash
# tools/shard_kstats.sh <n_lmks> <mode...>: rocprofv3 per-kernel averages of a rank's share of the headline graph (tools/shard_probe.py modes:
# engine = fused sweep + reduce, general = staged sweep + camera kernel, peer1 = the sharded loop with the peer-store exchange at one rank)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for mode in "$@"; do
  out=gpurun_out/shard_kstats/${L}_$mode; rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out -o run -- python tools/shard_probe.py --sizes $L --modes $mode --reps 160 --out $out/probe.json > $out/log.txt 2>&1
  grep "us/sweep" $out/log.txt
  python - $(find $out -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print(f"    {r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us  {float(r['Percentage']):5.1f} %")
PY
done
