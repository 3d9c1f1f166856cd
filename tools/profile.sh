#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel statistics + HBM counters of bench.py.
#   tools/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,fetch,write}/...csv; copy the summaries into profiles/.
# PMC passes are separate from the trace pass and from each other (FETCH_SIZE and WRITE_SIZE do not
# fit one pass: /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o run -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --single-batch "$@" > $OUT/bench_stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch -o run -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --single-batch "$@" > $OUT/bench_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/write -o run -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --single-batch "$@" > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | xargs ls -la
python tools/summarize_profile.py $OUT
