mkdir -p gpurun_out/r4h; O=gpurun_out/r4h
show() { python - "$@" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
f=lambda x: 0.0 if x is None else x*1e3
print(sys.argv[2], f"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel avg {f(r.get('kernel_avg_ms')):.1f} steady {f(r.get('kernel_steady_ms')):.1f} min {f(r.get('kernel_min_ms')):.1f} reduce {f(r.get('reduce_avg_ms')):.1f} are {d['are_after']:.6f}")
PY
}
GBP_HIP_LIB=$PWD/tools/libgbp_early.so python -m pytest tests/test_hip_parity.py tests/test_edge_shapes_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest_early.log 2>&1; tail -3 $O/pytest_early.log
for rep in 1 2 3; do
  for v in early default; do
    if [ $v = default ]; then unset GBP_HIP_LIB; else export GBP_HIP_LIB=$PWD/tools/libgbp_$v.so; fi
    python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b.json 2> $O/b.err; show $O/b.json "$v 1M"
  done
done
for v in early default; do
  if [ $v = default ]; then unset GBP_HIP_LIB; else export GBP_HIP_LIB=$PWD/tools/libgbp_$v.so; fi
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 --lmks 1000000 > $O/b.json 2> $O/b.err; show $O/b.json "$v 10M"
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --lmks 200000 > $O/b.json 2> $O/b.err; show $O/b.json "$v 2M"
  python tools/shard_probe.py --out $O/shards_$v.json --reps 100 --modes engine 2>&1 | grep "us/sweep"
  python bench.py --bal tests/golden/data/fr1desk.txt --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null > $O/b.json; show $O/b.json "$v fr1desk"
done
