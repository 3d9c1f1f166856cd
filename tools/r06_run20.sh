for v in default strided default strided; do
  if [ $v = default ]; then unset GBP_HIP_LIB; else export GBP_HIP_LIB=$PWD/tools/libgbp_$v.so; fi
  echo "== $v"; python tools/mode_probe.py 2>&1 | grep -E "^engine"
done
export GBP_HIP_LIB=$PWD/tools/libgbp_strided.so
python bench.py --lmks 200000 --steps 20 --warmup 5 --no-hbm-size 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('strided 2M are_check', d['cpu_baseline']['are_check'], d['ms_per_step'])"
for L in 1000000; do for v in default strided default strided; do
  if [ $v = default ]; then unset GBP_HIP_LIB; else export GBP_HIP_LIB=$PWD/tools/libgbp_$v.so; fi
  python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 --lmks $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v $L', f\"step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {(r['kernel_steady_ms'] or 0)*1e3:.1f} frac {r['frac']:.3f}\")"
done; done
