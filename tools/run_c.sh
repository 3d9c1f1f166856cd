mkdir -p gpurun_out/r4c; O=gpurun_out/r4c
for w in w12nr w8; do
  for rep in 1 2; do
  GBP_HIP_LIB=$PWD/tools/libgbp_$w.so python bench.py --no-cpu-baseline --steps 7 --warmup 0 > $O/bench_$w.json 2> $O/bench_$w.err
  python - $O/bench_$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[1], f"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {r['kernel_steady_ms']*1e3:.1f} min {r['kernel_min_ms']*1e3:.1f} reduce {r['reduce_avg_ms']*1e3:.1f} are {d['are_after']:.6f}")
PY
  done
done
PHASE_LIB=$PWD/tools/libgbp_phase12nr.so python tools/phase_profile.py > $O/phase_w12nr.txt 2>&1; head -16 $O/phase_w12nr.txt
