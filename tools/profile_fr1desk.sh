#!/bin/bash
# BASELINE configs 2-3 on the GPU box: kernel statistics + bench lines (with both CPU baselines) of the two fr1desk files.
#   tools/profile_fr1desk.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/prof_${TAG}_fr1desk
mkdir -p $OUT/summary
for f in fr1desk_small fr1desk; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_$f -o run -- python bench.py --bal tests/golden/data/$f.txt --steps 200 --warmup 20 --no-cpu-baseline --single-batch > $OUT/bench_stats_$f.log 2>&1
  cp $(find $OUT/stats_$f -name "*kernel_stats.csv" | head -1) $OUT/summary/${TAG}_${f}_kernel_stats.csv
  timeout 300 python bench.py --bal tests/golden/data/$f.txt --steps 200 --warmup 20 2>/dev/null >> $OUT/summary/${TAG}_fr1desk_bench.jsonl
done
head -4 $OUT/summary/*.csv; cut -c1-300 $OUT/summary/${TAG}_fr1desk_bench.jsonl
