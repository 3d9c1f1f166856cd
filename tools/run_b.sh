mkdir -p gpurun_out/r4b; O=gpurun_out/r4b
python -m pytest tests/test_compat_gpu.py tests/test_hip_parity.py -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for w in 8 12; do
  GBP_HIP_LIB=$PWD/tools/libgbp_w$w.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_w$w.json 2> $O/bench_w$w.err
  python - $O/bench_w$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[1], f"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {r['kernel_steady_ms']*1e3:.1f} relin {r.get('relinearising_sweeps',{}).get('kernel_avg_ms',0)*1e3:.1f} reduce {r['reduce_avg_ms']*1e3:.1f} are {d['are_after']:.6f}")
PY
  PHASE_LIB=$PWD/tools/libgbp_phase$w.so python tools/phase_profile.py > $O/phase_w$w.txt 2>&1; head -16 $O/phase_w$w.txt
done
