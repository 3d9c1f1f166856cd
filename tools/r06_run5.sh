mkdir -p gpurun_out/r06e
bash tools/ab_libs.sh 4 default geomlate > gpurun_out/r06e/ab_headline.txt 2>&1
for v in default geomlate default geomlate; do
  if [ $v = default ]; then unset GBP_HIP_LIB; else export GBP_HIP_LIB=$PWD/tools/libgbp_$v.so; fi
  echo "== $v" >> gpurun_out/r06e/ab_shards.txt
  timeout 600 python tools/shard_probe.py --sizes 50000 25000 12500 --modes engine --reps 160 --out gpurun_out/r06e/shard_$v.json 2>&1 | grep us/sweep >> gpurun_out/r06e/ab_shards.txt
  echo "== $v 2M" >> gpurun_out/r06e/ab_2m.txt
  python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 --lmks 200000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(f\"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {(r['kernel_steady_ms'] or 0)*1e3:.1f} frac {r['frac']:.3f}\")" >> gpurun_out/r06e/ab_2m.txt
done
cat gpurun_out/r06e/ab_headline.txt gpurun_out/r06e/ab_2m.txt gpurun_out/r06e/ab_shards.txt
