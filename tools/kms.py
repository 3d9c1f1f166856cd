import sys, json
for ln in sys.stdin:
    if ln.startswith('{"metric"'):
        d = json.loads(ln)
        print(f"kernel {d['roofline']['kernel_avg_ms']*1e3:.1f} us  step {d['ms_per_step']*1e3:.1f} us  {d['value']:.0f} it/s  are {d['are_after']:.4f}")
