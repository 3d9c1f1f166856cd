mkdir -p gpurun_out/r4d; O=gpurun_out/r4d
python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider -x > $O/pytest.log 2>&1; tail -12 $O/pytest.log
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[1], f"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {r['kernel_steady_ms']*1e3:.1f} relin {r.get('relinearising_sweeps',{}).get('kernel_avg_ms',0)*1e3:.1f} reduce {r['reduce_avg_ms']*1e3:.1f} are {d['are_after']:.6f}")
PY
