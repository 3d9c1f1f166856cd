mkdir -p gpurun_out/r4e; O=gpurun_out/r4e
python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/pytest.log 2>&1; tail -8 $O/pytest.log
python tools/shard_probe.py --out $O/shards.json --reps 200 2>&1 | grep "us/sweep"
for f in fr1desk_small fr1desk; do
  python bench.py --bal tests/golden/data/$f.txt --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null > $O/bench_$f.json
  python - $O/bench_$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[1], f"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} reduce {r['reduce_avg_ms']*1e3:.1f}")
PY
done
for a in "--no-fused" "--cams 2000" "--lmks 1000000 --steps 10 --warmup 3"; do
  python bench.py --no-cpu-baseline $a 2>/dev/null > $O/bench_x.json
  python - $O/bench_x.json "$a" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], f"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel {r['kernel']} avg {r['kernel_avg_ms']*1e3:.1f} steady {r.get('kernel_steady_ms',0)*1e3:.1f} frac {r['frac']:.3f}")
PY
done
