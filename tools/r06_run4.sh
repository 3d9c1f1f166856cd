mkdir -p gpurun_out/r06d
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06d/pytest.txt
timeout 600 python tools/shard_probe.py --sizes 50000 25000 12500 --modes engine peer1 --reps 240 --out gpurun_out/r06d/shard_probe.json > gpurun_out/r06d/shard_probe.txt 2>&1
bash tools/shard_kstats.sh 12500 engine peer1 > gpurun_out/r06d/kstats_12500.txt 2>&1
bash tools/shard_kstats.sh 25000 engine peer1 > gpurun_out/r06d/kstats_25000.txt 2>&1
for f in fr1desk.txt fr1desk_small.txt; do python bench.py --bal tests/golden/data/$f --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f', f\"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.2f} us kernel {r['kernel_avg_ms']*1e3:.2f} reduce {(r.get('reduce_avg_ms') or 0)*1e3:.2f}\")" >> gpurun_out/r06d/fr1desk.txt; done
tail -4 gpurun_out/r06d/pytest.txt; grep us/sweep gpurun_out/r06d/shard_probe.txt; cat gpurun_out/r06d/kstats_12500.txt gpurun_out/r06d/fr1desk.txt
