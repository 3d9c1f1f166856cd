#!/bin/bash
# Round 6: everything profiles/r06_* comes from, one gpurun call (tools/profile_round.sh does the counter passes and the default line)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
mkdir -p gpurun_out/r06_final
# a rank's share at 2 / 4 / 8 ranks: engine, whole tables, general sweep, with the (one-trip) peer exchange
timeout 900 python tools/shard_probe.py --sizes 50000 25000 12500 6250 --reps 240 --modes engine engine_whole general peer1 peer1_whole peer1g --out gpurun_out/r06_final/shard_probe.json > gpurun_out/r06_final/shard_probe.txt 2>&1
bash tools/shard_kstats.sh 12500 engine peer1 > gpurun_out/r06_final/shard_kstats_125k.txt 2>&1
bash tools/shard_kstats.sh 25000 engine peer1 > gpurun_out/r06_final/shard_kstats_250k.txt 2>&1
# BASELINE configs 2-3
for f in fr1desk.txt fr1desk_small.txt; do timeout 300 python bench.py --bal tests/golden/data/$f --steps 200 --warmup 20 > gpurun_out/r06_final/bench_$f.json 2> gpurun_out/r06_final/bench_$f.err; done
# what an N > 1 line carries (all ranks on this box's one GPU: field carriers, not measurements)
mkdir -p gpurun_out/r06_final/lines
GBP_KEEP_BENCH_LINES=$ROOT/gpurun_out/r06_final/lines timeout 1200 python -m pytest tests/test_peer_ipc_gpu.py -x -q -k "bench_n_ranks or rccl_that" 2>&1 | grep -E "passed|failed" > gpurun_out/r06_final/pytest_lines.txt
timeout 600 python tests/tools/g15b_trace.py > gpurun_out/r06_final/g15b.txt 2>&1
# other shapes of a million factors, camera windows, the general sweep
timeout 900 python tools/shape_sweep.py > gpurun_out/r06_final/shape_sweep.txt 2>&1
timeout 900 python tools/window_sweep.py > gpurun_out/r06_final/window_sweep.txt 2>&1
timeout 600 python tools/general_sweep.py > gpurun_out/r06_final/general_sweep.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/r06_final/pytest.txt
timeout 120 python __graft_entry__.py smoke > gpurun_out/r06_final/smoke.txt 2>&1
cat gpurun_out/r06_final/pytest.txt gpurun_out/r06_final/pytest_lines.txt; tail -2 gpurun_out/r06_final/smoke.txt; grep us/sweep gpurun_out/r06_final/shard_probe.txt
