mkdir -p gpurun_out/r4i; O=gpurun_out/r4i
show() { python - "$@" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
f=lambda x: 0.0 if x is None else x*1e3
print(sys.argv[2], f"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel avg {f(r.get('kernel_avg_ms')):.1f} steady {f(r.get('kernel_steady_ms')):.1f} min {f(r.get('kernel_min_ms')):.1f} reduce {f(r.get('reduce_avg_ms')):.1f} are {d['are_after']:.6f}")
PY
}
for rep in 1 2; do
  for v in w4 w6 default; do
    if [ $v = default ]; then unset GBP_HIP_LIB; else export GBP_HIP_LIB=$PWD/tools/libgbp_$v.so; fi
    python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b.json 2> $O/b.err; show $O/b.json "$v 1M"
  done
done
for v in w4 w6; do
  export GBP_HIP_LIB=$PWD/tools/libgbp_$v.so
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 --lmks 1000000 > $O/b.json 2> $O/b.err; show $O/b.json "$v 10M"
  python tools/shard_probe.py --out $O/shards_$v.json --reps 100 --modes engine --sizes 12500 2>&1 | grep "us/sweep"
done
