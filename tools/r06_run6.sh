mkdir -p gpurun_out/r06f
bash tools/ab_libs.sh 4 default eager > gpurun_out/r06f/ab_headline.txt 2>&1
for v in default eager default eager; do
  if [ $v = default ]; then unset GBP_HIP_LIB; else export GBP_HIP_LIB=$PWD/tools/libgbp_$v.so; fi
  echo "== $v" >> gpurun_out/r06f/ab_shards.txt
  timeout 600 python tools/shard_probe.py --sizes 50000 25000 12500 --modes engine --reps 160 --out gpurun_out/r06f/shard_$v.json 2>&1 | grep us/sweep >> gpurun_out/r06f/ab_shards.txt
  echo "== $v 2M" >> gpurun_out/r06f/ab_2m.txt
  python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 --lmks 200000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(f\"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {(r['kernel_steady_ms'] or 0)*1e3:.1f} frac {r['frac']:.3f}\")" >> gpurun_out/r06f/ab_2m.txt
done
unset GBP_HIP_LIB
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06f/pytest.txt
cat gpurun_out/r06f/ab_headline.txt gpurun_out/r06f/ab_2m.txt gpurun_out/r06f/ab_shards.txt gpurun_out/r06f/pytest.txt
