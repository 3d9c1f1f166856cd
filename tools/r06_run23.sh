mkdir -p gpurun_out/r06u
for L in 105000 110000 115000 120000 135000 150000 200000 300000 1000000; do
  python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 --lmks $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lmks $L', f\"step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {(r['kernel_steady_ms'] or 0)*1e3:.1f} ps/factor {(r['kernel_steady_ms'] or 0)*1e9/d['config']['n_factors']:.1f} frac {r['frac']:.3f}\")" | tee -a gpurun_out/r06u/size_sweep.txt
done
python tools/mode_probe.py 2>&1 | grep -E "^engine" | tee gpurun_out/r06u/mode_probe.txt
