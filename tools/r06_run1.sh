set -x
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06a/pytest.txt
timeout 120 python __graft_entry__.py smoke > gpurun_out/r06a/smoke.txt 2>&1
timeout 600 python tools/shard_probe.py --sizes 50000 25000 12500 --modes engine peer1 --reps 240 --out gpurun_out/r06a/shard_probe.json > gpurun_out/r06a/shard_probe.txt 2>&1
LMKS=12500 timeout 300 python tools/phase_profile.py > gpurun_out/r06a/phase_12500.txt 2>&1
LMKS=25000 timeout 300 python tools/phase_profile.py > gpurun_out/r06a/phase_25000.txt 2>&1
timeout 600 python tests/tools/g15b_trace.py > gpurun_out/r06a/g15b.txt 2>&1
timeout 600 python bench.py > gpurun_out/r06a/bench_default.json 2> gpurun_out/r06a/bench_default.err
tail -3 gpurun_out/r06a/pytest.txt
