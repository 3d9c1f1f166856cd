#!/bin/bash
set -u
mkdir -p gpurun_out/s1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s1/pytest.log
tail -25 gpurun_out/s1/pytest.log
timeout 600 python bench.py --no-cpu-baseline --dump-sweeps gpurun_out/s1/sweeps.npz > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err; echo "bench rc=$?"
python - <<'PY'
import numpy as np
d=np.load('gpurun_out/s1/sweeps.npz'); k=d['kernel_ms']; r=d['relin']
print('relin nonzero sweeps', np.nonzero(r)[0][:40], r[np.nonzero(r)[0][:40]])
print('kernel ms first 30', np.round(k[:30],4))
print('sorted top 12', np.round(np.sort(k)[-12:],4), 'mean', k.mean(), 'median', np.median(k))
PY
