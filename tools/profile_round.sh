#!/bin/bash
# Run on the GPU box (through gpurun): everything the committed profiles/<tag>_* files come from.
#   tools/profile_round.sh <tag>
#  1. rocprofv3 --kernel-trace --stats of the default bench (one batch) at 1M factors
#  2. separate PMC passes FETCH_SIZE / WRITE_SIZE (they do not fit one pass), 1M factors
#  3. the same three at 10M factors (--lmks 1000000): state far beyond the 256 MiB Infinity Cache
#  4. SQ occupancy / stall counters, one pass per group (each under timeout)
#  5. GBP_FUSED_DBG ablations (timing only; a scratch build with -DGBP_FUSED_DBG_SWITCHES)
#  6. the general sweep (--no-fused) kernel stats
# tools/summarize_round.py condenses the CSVs into gpurun_out/prof_<tag>/summary/ ; copy those files to profiles/.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --single-batch"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_1m -o run -- $B --steps 200 --warmup 20 > $OUT/bench_stats_1m.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -f csv -d $OUT/${c}_1m -o run -- $B --steps 10 --warmup 2 > $OUT/bench_${c}_1m.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_10m -o run -- $B --steps 40 --warmup 10 --lmks 1000000 > $OUT/bench_stats_10m.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -f csv -d $OUT/${c}_10m -o run -- $B --steps 6 --warmup 2 --lmks 1000000 > $OUT/bench_${c}_10m.log 2>&1
done
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp -f csv -d $OUT/sq_$i -o run -- $B --steps 6 --warmup 2 > $OUT/bench_sq_$i.log 2>&1
  echo "sq group $i ($grp) rc=$?"
done
# the ablation switches are compiled out of the product kernel (run-time tests cost it 1.5 us per sweep): a scratch copy of the library has them
python -m gbp_amd.build --out "$ROOT/tools/libgbp_dbg.so" -DGBP_FUSED_DBG_SWITCHES > /dev/null
for dbg in 0 1 4 5 8 12 14; do
  GBP_HIP_LIB="$ROOT/tools/libgbp_dbg.so" GBP_FUSED_DBG=$dbg timeout 300 $B --steps 200 --warmup 20 > $OUT/bench_dbg$dbg.json 2> $OUT/bench_dbg$dbg.err
done
rm -f "$ROOT/tools/libgbp_dbg.so"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_general -o run -- $B --steps 100 --warmup 10 --no-fused > $OUT/bench_stats_general.log 2>&1
python tools/summarize_round.py $OUT $TAG
# bench.py reports `traffic` only from a profiles/<tag>_hbm_traffic.json that names the library it runs: put this call's counter passes
# there first, then take the lines that are kept (the 1M run under rocprofv3 once more, the default command)
cp $OUT/summary/${TAG}_hbm_traffic.json profiles/${TAG}_hbm_traffic.json
rm -rf $OUT/stats_1m
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_1m -o run -- $B --steps 200 --warmup 20 > $OUT/bench_stats_1m.log 2>&1
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python tools/summarize_round.py $OUT $TAG
