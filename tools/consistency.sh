# tools/consistency.sh <tag>: does the instrumented replay agree with the timed batches?  (kernel + reduce vs step, three short runs + one long)
O=gpurun_out/${1:-cons}; mkdir -p $O
show() { python - "$@" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
f=lambda x: 0.0 if x is None else x*1e3
print(sys.argv[2], f"{d['value']:.0f}", f"step {d['ms_per_step']*1e3:.2f} kernel {f(r.get('kernel_avg_ms')):.2f} reduce {f(r.get('reduce_avg_ms')):.2f} sum {f(r.get('kernel_avg_ms'))+f(r.get('reduce_avg_ms')):.2f} consistent {r.get('consistent')} events {f(r.get('kernel_event_ms')):.2f} frac {r.get('frac'):.3f} hbm {r.get('frac_hbm_bound')} traffic {r.get('traffic')}")
PY
}
for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null > $O/s20_$i.json; show $O/s20_$i.json s20; done
python bench.py --no-cpu-baseline --no-hbm-size 2>/dev/null > $O/default.json; show $O/default.json default
