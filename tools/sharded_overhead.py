#!/usr/bin/env python3
"""Per-sweep cost of the landmark-sharded driver at world size 1 (RCCL backend, one GPU): what the Python loop + the
collective + the extra launches add over the single-GPU engine.  torchrun-less: sets up a 1-rank process group itself."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import torch, torch.distributed as dist
from gbp_amd.synthetic import make_synthetic
from gbp_amd.sharded import ShardedBA
from gbp_amd.engine import BAEngine
n_l = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
p = make_synthetic(n_cams=500, n_lmks=n_l, obs_per_lmk=10, seed=0)
for name, g in (('ShardedBA(world=1)', ShardedBA(p, device=0)), ('BAEngine', BAEngine.from_problem(p))):
    g.generate_priors_var(50.0); g.update_all_beliefs(); g.iterate(20); g.sync()
    t0 = time.perf_counter(); g.iterate(200); t1 = time.perf_counter(); g.sync(); t2 = time.perf_counter()
    print(f"{name:20s} F={p.n_factors}: enqueue {1e6 * (t1 - t0) / 200:.1f} us/sweep (CPU), done {1e6 * (t2 - t0) / 200:.1f} us/sweep")
dist.destroy_process_group()
