mkdir -p gpurun_out/r06c
timeout 300 python -m pytest tests/test_peer_ipc_gpu.py -x -q -k "rccl_that_does_not" 2>&1 | tail -40 > gpurun_out/r06c/pytest_rccl.txt
timeout 900 python -m pytest tests/test_peer_ipc_gpu.py tests/test_sharded_gpu.py tests/test_sharded_fuzz_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r06c/pytest_peer.txt
bash tools/ab_libs.sh 2 default pf1 pf2 pf3 > gpurun_out/r06c/ab_headline.txt 2>&1
timeout 600 python tools/shard_probe.py --sizes 50000 25000 12500 --modes engine peer1 --reps 240 --out gpurun_out/r06c/shard_probe.json > gpurun_out/r06c/shard_probe.txt 2>&1
cat gpurun_out/r06c/pytest_rccl.txt | tail -30; cat gpurun_out/r06c/pytest_peer.txt | tail -5; cat gpurun_out/r06c/ab_headline.txt; grep us/sweep gpurun_out/r06c/shard_probe.txt
