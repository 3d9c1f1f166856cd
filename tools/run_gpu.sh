# tools/run_gpu.sh <tag> : the GPU suite + the bench lines of the general sweep (scratch output under gpurun_out/<tag>)
O=gpurun_out/${1:-run}; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1; tail -6 $O/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr"
show() { python - "$@" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
f=lambda x: 0.0 if x is None else x*1e3
print(sys.argv[2], f"{d['value']:.0f} it/s step {d['ms_per_step']*1e3:.1f} us kernel {r.get('kernel')} avg {f(r.get('kernel_avg_ms')):.1f} steady {f(r.get('kernel_steady_ms')):.1f} reduce {f(r.get('reduce_avg_ms')):.1f} frac {r.get('frac'):.3f} are {d['are_after']:.6f}")
PY
}
python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 2>/dev/null > $O/b.json; show $O/b.json fused
python bench.py --no-cpu-baseline --no-fused --steps 20 --warmup 5 2>/dev/null > $O/b.json; show $O/b.json nofused
GBP_TILE_KERNEL=1 python bench.py --no-cpu-baseline --no-fused --steps 20 --warmup 5 2>/dev/null > $O/b.json; show $O/b.json nofused-tile-kernel
python bench.py --no-cpu-baseline --cams 2000 --steps 20 --warmup 5 2>/dev/null > $O/b.json; show $O/b.json cams2000
python bench.py --no-cpu-baseline --cams 1000 --steps 20 --warmup 5 2>/dev/null > $O/b.json; show $O/b.json cams1000
python bench.py --no-cpu-baseline --cams 3000 --steps 20 --warmup 5 2>/dev/null > $O/b.json; show $O/b.json cams3000
