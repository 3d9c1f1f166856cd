#!/usr/bin/env python3
"""Per-sweep cost of the landmark-sharded driver at world size 1 (RCCL backend, one GPU) against the plain engine:
the in-library loop (gbp_ba_iterate_sharded, ncclAllGather on the library's own communicator, forced although there is one
rank), the Python-driven loop (shard_begin / all_gather_into_tensor / shard_end) and the sharded path as it runs at N = 1
(no exchange).  torchrun-less: sets up a 1-rank process group itself.

    python tools/sharded_overhead.py [n_lmks ...]        # default 100000 (1M factors) and 12500 (the per-rank size at 8 GPUs)
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import torch, torch.distributed as dist
from gbp_amd.synthetic import make_synthetic
from gbp_amd.sharded import ShardedBA
from gbp_amd.engine import BAEngine
sizes = [int(a) for a in sys.argv[1:]] or [100_000, 12_500]
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
for n_l in sizes:
    p = make_synthetic(n_cams=500, n_lmks=n_l, obs_per_lmk=10, seed=0)
    for name, mk in (('BAEngine', lambda: BAEngine.from_problem(p)),
                     ('sharded N=1 (no exchange)', lambda: ShardedBA(p, device=0)),
                     ('sharded in-library + RCCL', lambda: ShardedBA(p, device=0, always_exchange=True)),
                     ('sharded python loop + RCCL', lambda: ShardedBA(p, device=0, library_loop=False))):
        g = mk()
        g.generate_priors_var(50.0); g.update_all_beliefs(); g.iterate(20); g.sync()
        t0 = time.perf_counter(); g.iterate(200); t1 = time.perf_counter(); g.sync(); t2 = time.perf_counter()
        print(f"F={p.n_factors:8d} {name:28s}: enqueue {1e6 * (t1 - t0) / 200:6.1f} us/sweep (host), done {1e6 * (t2 - t0) / 200:6.1f} us/sweep", flush=True)
        g.close()
dist.destroy_process_group()
