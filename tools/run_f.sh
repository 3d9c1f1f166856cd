mkdir -p gpurun_out/r4f; O=gpurun_out/r4f
python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
show() { python - "$@" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
f=lambda x: 0.0 if x is None else x*1e3
print(sys.argv[2], f"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel {r.get('kernel')} avg {f(r.get('kernel_avg_ms')):.1f} steady {f(r.get('kernel_steady_ms')):.1f} reduce {f(r.get('reduce_avg_ms')):.1f} frac {r.get('frac')}")
PY
}
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; show $O/bench.json default
python tools/shard_probe.py --out $O/shards.json --reps 200 --modes engine peer1 2>&1 | grep "us/sweep"
for f in fr1desk_small fr1desk; do
  python bench.py --bal tests/golden/data/$f.txt --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null > $O/bench_$f.json; show $O/bench_$f.json $f
done
python bench.py --no-cpu-baseline --no-fused 2>/dev/null > $O/bench_nofused.json; show $O/bench_nofused.json nofused
python bench.py --no-cpu-baseline --cams 2000 2>/dev/null > $O/bench_c2000.json; show $O/bench_c2000.json cams2000
python bench.py --no-cpu-baseline --cams 1000 2>/dev/null > $O/bench_c1000.json; show $O/bench_c1000.json cams1000
