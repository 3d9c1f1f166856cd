#!/usr/bin/env python3
"""Run the fuzz comparison of tests/test_fuzz_gpu.py over a range of seeds (GPU box): python tests/tools/fuzz_many.py 24 400"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..')); sys.path.insert(0, os.path.join(HERE, '..'))
import test_fuzz_gpu as tf
from oracle import oracle as om
om.build()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, total = [], 0
for seed in range(lo, hi):
    try:
        tf.test_random_shapes_against_oracle.__wrapped__(om, seed) if hasattr(tf.test_random_shapes_against_oracle, '__wrapped__') else tf.test_random_shapes_against_oracle(om, seed)
        total += tf.COMPARED.get(seed, 0)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
print(f"seeds {lo}..{hi - 1}: {len(bad)} failures, {total} sweeps compared")
for b in bad[:20]:
    print("  ", b)
