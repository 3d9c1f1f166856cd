#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (tests/test_bench_launch.py): bench.py's launch path in the build container, where there is no GPU.

bench.py itself cannot be told to time anything but the HIP engine.  This wrapper imports it and hands its main() a double for a
rank's engine -- tests/tools/shard_double.py, the C oracle behind the sharded driver's host logic -- so that spawning the ranks, the
rendezvous on 127.0.0.1 and the one JSON line from rank 0 can be exercised on CPU over gloo.  The line is marked "dry_run".
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(HERE, '..'))

import bench                                   # noqa: E402
from tools.shard_double import factory         # noqa: E402

if __name__ == '__main__':
    bench.main(shard_factory=factory, script=os.path.abspath(__file__))
