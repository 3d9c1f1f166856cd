#!/bin/bash
set -u
mkdir -p gpurun_out/s3
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/s3/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s3/pytest.log
tail -30 gpurun_out/s3/pytest.log
timeout 600 python tools/setup_time.py 2>&1 | tee gpurun_out/s3/setup.log
