#!/bin/bash
set -u
mkdir -p gpurun_out/s2
timeout 600 python bench.py --no-cpu-baseline --single-batch --dump-sweeps gpurun_out/s2/sweeps_a.npz > gpurun_out/s2/bench_a.json 2> gpurun_out/s2/bench_a.err; echo "bench rc=$?"
timeout 600 python bench.py --no-cpu-baseline --single-batch --dump-sweeps gpurun_out/s2/sweeps_b.npz > gpurun_out/s2/bench_b.json 2> gpurun_out/s2/bench_b.err; echo "bench rc=$?"
python - <<'PY'
import numpy as np
a=np.load('gpurun_out/s2/sweeps_a.npz'); b=np.load('gpurun_out/s2/sweeps_b.npz')
print(np.round(a['kernel_ms'][100:140],3)); print(np.round(b['kernel_ms'][100:140],3)); print(a['relin'][100:140])
print(a['batch_s'], b['batch_s'])
PY
