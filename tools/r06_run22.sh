mkdir -p gpurun_out/r06t
for L in 115000 135000 200000 300000 1000000; do for mib in 200 220 240 256 280; do
  export GBP_FUSED_PIN_MIB=$mib
  python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 --lmks $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lmks $L keep $mib', f\"step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {(r['kernel_steady_ms'] or 0)*1e3:.1f} frac {r['frac']:.3f}\")" | tee -a gpurun_out/r06t/keep_sweep.txt
done; done
