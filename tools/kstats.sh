#!/bin/bash
# tools/kstats.sh [bench args]  -- rocprofv3 kernel statistics (top lines) of one bench batch
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rm -rf gpurun_out/kstats; mkdir -p gpurun_out/kstats
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/kstats -o run -- python bench.py --no-cpu-baseline --single-batch "$@" > gpurun_out/kstats/bench.log 2>&1
head -4 $(find gpurun_out/kstats -name "*kernel_stats.csv" | head -1) | cut -c1-140
