mkdir -p gpurun_out/r06r
timeout 900 python tools/shard_probe.py --sizes 50000 25000 12500 6250 --modes engine engine_whole general peer1 peer1_whole peer1g --reps 240 --out gpurun_out/r06r/shard_probe.json 2>&1 | grep us/sweep > gpurun_out/r06r/shard_probe.txt
cat gpurun_out/r06r/shard_probe.txt
