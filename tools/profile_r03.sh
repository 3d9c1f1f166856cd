#!/bin/bash
# Round 3's committed profiles: tools/profile_round.sh + the fr1desk files + shard probe (plain engine / general sweep / sharded loop
# with the peer-store exchange, merged and split launches) + the linear engine + the size sweep around the Infinity Cache cliff.
#   tools/profile_r03.sh [tag]
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
bash tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1
bash tools/profile_fr1desk.sh $TAG > gpurun_out/profile_fr1desk_$TAG.log 2>&1
OUT=gpurun_out/prof_$TAG; S=$OUT/summary; mkdir -p $S
cp gpurun_out/prof_${TAG}_fr1desk/summary/* $S/ 2>/dev/null
python tools/shard_probe.py --modes engine general peer1 --out $S/${TAG}_shards.json > $OUT/shards.log 2>&1
GBP_PEER_SPLIT=1 python tools/shard_probe.py --modes peer1 --out $S/${TAG}_shards_split.json >> $OUT/shards.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for d in 3 6; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/lin$d -o run -- python tools/bench_linear.py --dofs $d --no-cpu-baseline > $OUT/lin$d.log 2>&1
  cp $(find $OUT/lin$d -name "*kernel_stats.csv" | head -1) $S/${TAG}_linear_d${d}_kernel_stats.csv
  timeout 300 python tools/bench_linear.py --dofs $d 2>/dev/null >> $S/${TAG}_linear_bench.jsonl
done
for n in 50000 70000 85000 100000 105000 110000 120000 150000 200000; do
  python bench.py --no-cpu-baseline --lmks $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(json.dumps({'n_factors': d['config']['n_factors'], 'us_per_step': d['ms_per_step']*1e3, 'kernel_us': r['kernel_avg_ms']*1e3, 'ns_per_factor': r['kernel_avg_ms']*1e9/d['config']['n_factors'], 'layout_MB': r['bytes_per_launch']/1e6, 'frac': r['frac']}))" >> $S/${TAG}_size_sweep.jsonl
done
ls -la $S
