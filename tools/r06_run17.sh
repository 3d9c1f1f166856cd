mkdir -p gpurun_out/r06q
PEER=1 python tools/boundary_probe.py 12500 25000 2>&1 | grep -v amdgpu > gpurun_out/r06q/boundary_peer.txt
timeout 600 python tools/shard_probe.py --sizes 50000 25000 12500 6250 --modes engine peer1 peer1g --reps 240 --out gpurun_out/r06q/shard_probe.json 2>&1 | grep us/sweep > gpurun_out/r06q/shard_probe.txt
timeout 1200 python -m pytest tests/test_peer_ipc_gpu.py tests/test_sharded_gpu.py tests/test_sharded_fuzz_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/r06q/pytest_peer.txt
cat gpurun_out/r06q/boundary_peer.txt gpurun_out/r06q/shard_probe.txt gpurun_out/r06q/pytest_peer.txt
