mkdir -p gpurun_out/r06s
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/r06s/pytest.txt
cat gpurun_out/r06s/pytest.txt
# the share kept cacheable, with the strided walk: 2M, 1.35M and 10M factors
for L in 200000 135000; do for mib in default 100 140 180 220; do
  if [ $mib = default ]; then unset GBP_FUSED_PIN_MIB; else export GBP_FUSED_PIN_MIB=$mib; fi
  python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 --lmks $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lmks $L keep $mib', f\"step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {(r['kernel_steady_ms'] or 0)*1e3:.1f} frac {r['frac']:.3f}\")" | tee -a gpurun_out/r06s/keep_sweep.txt
done; done
unset GBP_FUSED_PIN_MIB
for L in 110000 115000 120000 150000 300000; do
  python bench.py --no-cpu-baseline --no-hbm-size --steps 20 --warmup 5 --lmks $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lmks $L', f\"step {d['ms_per_step']*1e3:.1f} us kernel avg {r['kernel_avg_ms']*1e3:.1f} steady {(r['kernel_steady_ms'] or 0)*1e3:.1f} ps/factor {(r['kernel_steady_ms'] or 0)*1e9/d['config']['n_factors']:.1f} frac {r['frac']:.3f}\")" | tee -a gpurun_out/r06s/size_sweep.txt
done
