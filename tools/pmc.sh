#!/bin/bash
# tools/pmc.sh <tag> "<counters>" [bench args]  -- one rocprofv3 PMC pass over a short bench run
set -u
TAG=$1; CTR=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout 200 rocprofv3 --pmc $CTR -f csv -d $OUT -o run -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --single-batch "$@" > $OUT/bench.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row['Kernel_Name'].split('(')[0][:40], row['Counter_Name'])
        agg[k][0] += 1; agg[k][1] += float(row['Counter_Value'])
for (k, c), (n, s) in sorted(agg.items()):
    if 'sweep' in k or 'k_factor<' in k or 'reduce' in k:
        print(f'{k:40s} {c:24s} n={n:3d} mean={s/n:.6g}')
PY
