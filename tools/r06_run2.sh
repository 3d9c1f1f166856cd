# A/B of the L2 touch-prefetch (GBP_PF_DIST) against the product build, alternating, one call
mkdir -p gpurun_out/r06b
bash tools/ab_libs.sh 3 default pf4 pf8 pf16 > gpurun_out/r06b/ab_headline.txt 2>&1
for v in default pf8 pf16; do
  if [ $v = default ]; then unset GBP_HIP_LIB; else export GBP_HIP_LIB=$PWD/tools/libgbp_$v.so; fi
  echo "== $v" >> gpurun_out/r06b/ab_shards.txt
  timeout 600 python tools/shard_probe.py --sizes 50000 25000 12500 --modes engine peer1 --reps 160 --out gpurun_out/r06b/shard_$v.json >> gpurun_out/r06b/ab_shards.txt 2>&1
  echo "== $v 2M" >> gpurun_out/r06b/ab_2m.txt
  python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 --lmks 200000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(f\"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {(r['kernel_steady_ms'] or 0)*1e3:.1f} frac {r['frac']:.3f}\")" >> gpurun_out/r06b/ab_2m.txt
done
unset GBP_HIP_LIB
timeout 300 python -m pytest tests/test_peer_ipc_gpu.py -x -q -k "rccl_that_does_not" 2>&1 | tail -3 > gpurun_out/r06b/pytest_rccl.txt
cat gpurun_out/r06b/ab_headline.txt gpurun_out/r06b/ab_2m.txt
