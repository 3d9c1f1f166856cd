#!/usr/bin/env python3
"""bench.py -- GBP iterations/s on the synthetic 500-cam x 100k-landmark x 1M-factor BA graph.

    python bench.py --gpus N --steps K --warmup W

A "step" is one FactorGraph.synchronous_iteration(robustify=True, local_relin=True) (gbp/gbp.py:86-92) over the whole
graph, inputs resident in HBM.  N > 1 shards the graph by landmark across N ranks, one process per GPU, one RCCL
all-gather of the camera partial sums per iteration inside the library (gbp_ba_iterate_sharded); the total work is
fixed, so scaling is "strong".  Started without torchrun, `--gpus N` spawns its own N ranks (re-exec under
torch.distributed.run on 127.0.0.1); started under torchrun it uses the ranks it is given.  Rank 0 prints ONE JSON line.

Timing protocol.  Every batch starts from the same state -- the graph right after generate_priors_var +
update_all_beliefs, restored from a device-resident checkpoint -- runs W untimed sweeps and then EXACTLY K timed sweeps bracketed by a
barrier + device synchronisation on both sides (max over ranks).  Batches are repeated until >= 0.5 s have been timed;
`value` = K / median batch time, the minimum is reported beside it.  HIP events bracket every launch of the dominant
kernel in one extra, untimed replay of the same batch, and the device counts the factors that relinearise in each sweep:
steady sweeps (fewer than 1 factor in 1000 relinearises: the case SURVEY 8d's byte count describes) and relinearising
sweeps are reported separately.

Roofline bookkeeping.  `roofline.achieved` = the bytes the engine's data layout MUST move per launch of the dominant
kernel (F (21 read + 10 written doubles + 12 B of indices / state) + L (24 + 12 doubles) + one
camera table per workgroup) / the mean steady launch time; `frac` = achieved / 8 TB/s, never above 1.  `traffic` = HBM
bytes per launch measured by the PMC passes committed under profiles/.  The survey's 1072 B/factor model of a dense
two-pass implementation is kept only as `survey_equivalent_*`: this engine does that work in fewer bytes.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md:35
MIN_TIMED_S = 0.5
MAX_BATCHES = 2000          # (a 20-step batch of a 15 us sweep is 0.3 ms: the cap only bounds degenerate cases)


def survey_bytes(F, L, C):
    """fp64, packed-symmetric dense messages, two passes, steady-state iteration."""
    return F * 1072 + L * 168 + C * 480


def layout_bytes(F, L, C, n_blocks, fused):
    """Bytes one launch of the dominant kernel must move in THIS engine's layout."""
    if fused:       # k_sweep_wat: x0 9 z 2 | msgs 10 in, 10 out | meta 4 B, state 4 B in + 4 B out; landmark record 24 in, belief+mean 12 out; tables out
        return F * ((21 + 10) * 8 + 12) + L * (24 + 12) * 8 + n_blocks * C * 28 * 8      # (table rows: 27 sums + 1 pad double)
    # k_factor_tile: the same per-factor / per-landmark streams + what rebuilds the camera message, staged camera-major
    # (x0 9 | q_C 2 | W 3 = 14 doubles, + cpos)
    return F * ((21 + 10) * 8 + 12 + 14 * 8 + 4) + L * (24 + 12) * 8


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(problem, budget_s=20.0):
    """Time the CPU oracle (oracle/gbp_oracle.c, a port of the reference's algorithm) on the same workload,
    bounded to ~budget_s of wall time: a few whole sweeps of the same graph at 1 thread and at several OpenMP widths."""
    from oracle import oracle
    o = oracle.OracleBA.from_problem(problem, threads=1)
    o.generate_priors_var(50.0)
    o.update_all_beliefs()
    t0 = time.perf_counter()
    o.iterate(1)                                              # also the warm-up sweep
    dt1 = time.perf_counter() - t0
    best = (dt1, 1, 1)
    spent = dt1
    for threads in sorted({min(host_cores(), t) for t in (8, 32, 64, host_cores())}):
        if threads == 1 or spent > budget_s:
            continue
        o.set_threads(threads)
        o.iterate(1)
        n = int(max(1, min(10, (budget_s - spent) / 4 // max(best[0], 1e-3))))
        t0 = time.perf_counter()
        o.iterate(n)
        dt = (time.perf_counter() - t0) / n
        spent += dt * (n + 1)
        if dt < best[0]:
            best = (dt, threads, n)
    return {"value": 1.0 / best[0], "unit": "iter/s", "cores": best[1], "kind": "port",
            "sample": f"{best[2]} whole sweeps of the same {problem.n_factors}-factor graph (after warm-up), C oracle with "
                      f"OpenMP over factors/variables, best of several thread counts on {host_cores()} usable cores",
            "value_1thread": 1.0 / dt1, "us_per_factor_iter_1thread": dt1 / problem.n_factors * 1e6}


def cpu_baseline_numpy(problem, budget_s=15.0):
    """the numpy restatement with the reference's cost model (one Python object per factor, ~40 numpy calls per
    factor and sweep) at 1 thread -- only sensible on the small graphs of BASELINE configs 2-3."""
    from oracle.numpy_ba import NumpyBA
    n = NumpyBA(problem)
    n.generate_priors_var(50.0)
    n.update_all_beliefs()
    t0 = time.perf_counter()
    n.iterate(1)
    dt = time.perf_counter() - t0
    k = int(max(0, min(5, (budget_s - dt) // max(dt, 1e-3))))
    if k:
        t0 = time.perf_counter()
        n.iterate(k)
        dt = (time.perf_counter() - t0) / k
    return {"value": 1.0 / dt, "unit": "iter/s", "cores": 1, "kind": "port",
            "sample": f"{max(k, 1)} whole sweeps of the same {problem.n_factors}-factor graph, object-per-factor numpy graph (oracle/numpy_ba.py)",
            "us_per_factor_iter": dt / problem.n_factors * 1e6}


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/rNN_hbm_traffic.json)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_hbm_traffic.json')))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])).get('traffic_bytes_per_launch'), os.path.basename(files[-1])
    except (OSError, ValueError):
        return None, None


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: become the launcher of N ranks of this very command line."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, host_cores() // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--cams', type=int, default=500)
    ap.add_argument('--lmks', type=int, default=100_000)
    ap.add_argument('--obs', type=int, default=10)
    ap.add_argument('--bal', default=None, help='BAL text file instead of the synthetic graph (BASELINE configs 2-3, e.g. '
                                                'tests/golden/data/fr1desk.txt); not the headline workload')
    ap.add_argument('--no-fused', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--single-batch', action='store_true', help='one timed batch only (profiling runs)')
    ap.add_argument('--sharded', action='store_true', help='take the N > 1 code path (process group, ShardedBA, RCCL exchange forced) even with one rank')
    ap.add_argument('--python-loop', action='store_true', help='N > 1: drive the sweeps from Python (shard_begin / all_gather / shard_end)')
    ap.add_argument('--dump-sweeps', default=None, help='write the per-sweep kernel times (ms) and relinearisation counts of the instrumented replay to this .npz')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend of the side channel (tests: gloo)')
    ap.add_argument('--engine-factory', default=None,
                    help='module:callable building a rank\'s engine double (launch-path tests in the build container; the result '
                         'is then marked "dry_run" and is not a measurement)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    # Only the JSON line may reach stdout: native libraries (RCCL prints a version banner at communicator creation) write to
    # file descriptor 1 directly, so the descriptor itself is pointed at stderr and the result goes out through a saved copy.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dry = args.engine_factory is not None
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a MI355X: the HIP engine has no CPU fallback")
        torch.cuda.set_device(local_rank)

    if args.bal:
        from gbp_amd.balio import read_bal
        problem = read_bal(args.bal)
        workload = f"BAL file {os.path.basename(args.bal)}"
    else:
        from gbp_amd.synthetic import make_synthetic
        problem = make_synthetic(n_cams=args.cams, n_lmks=args.lmks, obs_per_lmk=args.obs, seed=0)
        workload = "synthetic BAL"
    F, L, C = problem.n_factors, problem.n_lmks, problem.n_cams

    # one node: the collectives' bootstrap sockets need no NIC and no resolvable hostname (the data path is xGMI either way)
    os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
    dist = None
    if world > 1 or dry or args.sharded:
        import torch.distributed as dist
        from gbp_amd.sharded import ShardedBA
        if dry:
            mod, fn = args.engine_factory.split(':')
            sys.path.insert(0, os.path.join(REPO, 'tests'))
            dist.init_process_group(args.backend)
            graph = ShardedBA(problem, engine_factory=getattr(importlib.import_module(mod), fn))
        else:
            if 'MASTER_ADDR' not in os.environ:                     # --sharded without torchrun: a one-rank group of our own
                os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
            dist.init_process_group(args.backend, device_id=torch.device('cuda', local_rank))
            graph = ShardedBA(problem, device=local_rank, fused=not args.no_fused, library_loop=not args.python_loop,
                              always_exchange=args.sharded)
    else:
        from gbp_amd.engine import BAEngine
        graph = BAEngine.from_problem(problem, device=local_rank, fused=not args.no_fused)

    def fence():
        if dist is not None:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    graph.generate_priors_var(50.0)
    graph.update_all_beliefs()
    graph.sync()
    if not dry:
        graph.snapshot_state()                                 # device-resident: restoring it leaves no idle gap before the sweeps

    def batch(timing=False):
        """Restore the initial state, W untimed sweeps, then K sweeps between two fences.  Returns wall seconds (max over ranks)."""
        if not dry:
            graph.restore_snapshot()
        graph.iterate(args.warmup)
        graph.sync()
        if timing:
            graph.set_kernel_timing(1)
        fence()
        t0 = time.perf_counter()
        graph.iterate(args.steps)
        graph.sync()
        fence()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device='cpu' if dry else 'cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    batch()                                                   # untimed: page in code, caches, clocks
    times = [batch()]
    while not args.single_batch and sum(times) < MIN_TIMED_S and len(times) < MAX_BATCHES:
        times.append(batch())
    times = np.array(times)
    dt_med, dt_min = float(np.median(times)), float(times.min())

    # kernel-level picture from one more replay of the same batch with HIP events around every launch of the dominant kernel
    k_times, relin, k_name = np.zeros(0), np.zeros(0, np.int64), ''
    if not dry:
        dt_instr = batch(timing=True)
        k_times = graph.kernel_times()                         # ms, one per sweep of the timed region
        _, _, k_name = graph.kernel_timing()
        graph.set_kernel_timing(0)
        if args.steps <= 512:
            relin = np.asarray(graph.relin_counts(args.steps), dtype=np.int64)
    are = graph.are()
    if args.dump_sweeps and rank == 0:
        np.savez(args.dump_sweeps, kernel_ms=k_times, relin=relin, batch_s=times)

    if rank == 0:
        info = dict(fused=False, n_blocks=0) if dry else graph.info()
        its = args.steps / dt_med
        F_local, L_local = graph.F, graph.L
        fused = bool(info.get('fused'))
        lay = layout_bytes(F_local, L_local, C, info.get('n_blocks', 0), fused)
        # steady = (almost) nobody relinearises: fewer than 1 factor in 1000 (they add < 0.1 % of the sweep's bytes)
        steady = relin * 1000 < F if relin.size == k_times.size else np.ones(k_times.size, bool)
        if not steady.any():
            steady = np.ones(k_times.size, bool)
        k_steady = float(k_times[steady].mean()) if k_times.size else 0.0
        achieved = lay / (k_steady * 1e-3) / 1e9 if k_steady else 0.0
        traffic, traffic_src = measured_traffic() if (world == 1 and fused and F == 1_000_000 and not dry) else (None, None)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "kernel": k_name, "bytes_per_launch": lay,
                "bytes_model": "engine layout, DESIGN.md section 4: F*(31 doubles + 12 B) + L*36 doubles + camera tables",
                "kernel_avg_ms": k_steady, "kernel_median_ms": float(np.median(k_times[steady])) if k_times.size else 0.0,
                "kernel_min_ms": float(k_times[steady].min()) if k_times.size else 0.0,
                "kernel_launches_timed": int(steady.sum()), "kernel_timing": "HIP events around every launch, separate replay of the batch",
                "survey_equivalent_bytes": survey_bytes(F_local, L_local, C),
                "survey_equivalent_gbs": survey_bytes(F_local, L_local, C) / (k_steady * 1e-3) / 1e9 if k_steady else 0.0}
        if traffic and k_steady:
            roof["traffic_source"] = f"profiles/{traffic_src}"
            roof["traffic_gbs"] = traffic / (k_steady * 1e-3) / 1e9
            roof["traffic_frac"] = roof["traffic_gbs"] / HBM_PEAK_GBS
        if k_times.size and (~steady).any():
            full = relin * 2 > F
            roof["relinearising_sweeps"] = {"count": int((~steady).sum()), "kernel_avg_ms": float(k_times[~steady].mean()),
                                            "kernel_max_ms": float(k_times[~steady].max()),
                                            "factors_per_sweep_mean": float(relin[~steady].mean()), "factors_per_sweep_max": int(relin.max()),
                                            "extra_bytes_per_factor": 72}
            if full.any():
                roof["relinearising_sweeps"]["all_factors"] = {"count": int(full.sum()), "kernel_avg_ms": float(k_times[full].mean())}
        out = {
            "metric": "GBP iterations/sec (whole node), 1M-factor BA graph" if F == 1_000_000 else f"GBP iterations/sec, {F}-factor BA graph",
            "value": its, "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_med / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "fixture file (tests/golden/data)" if args.bal else "synthetic",
            "config": {"workload": f"{workload} {C} cams x {L} landmarks x {F} reprojection factors "
                                   + ("" if args.bal else "(gbp_amd.synthetic.make_synthetic seed 0), ") + "ba.py defaults, loss=None",
                       "n_cams": C, "n_lmks": L, "n_factors": F,
                       "parallelism": f"landmark-sharded x{world}, RCCL all-gather of camera partial sums per sweep" if world > 1 else "single GPU",
                       "sweep": "fused" if fused else "general",
                       "loop": ("python" if (args.python_loop or dry) else "in-library") if (world > 1 or args.sharded) else "gbp_ba_iterate"},
            "timing": {"protocol": "state restored before every batch; W warm-up + K timed sweeps between barrier+synchronize; median over batches",
                       "batches": int(times.size), "timed_seconds": float(times.sum()),
                       "ms_per_step_median": dt_med / args.steps * 1e3, "ms_per_step_min": dt_min / args.steps * 1e3,
                       "ms_per_step_first": float(times[0]) / args.steps * 1e3},
            "roofline": roof,
            "are_after": are,
        }
        if dry:
            out["dry_run"] = True
        if world == 1 and not args.no_cpu_baseline and not dry:
            out["cpu_baseline"] = cpu_baseline(problem)
            if args.bal and F <= 50_000:
                out["cpu_baseline_numpy"] = cpu_baseline_numpy(problem)
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist is not None:
        dist.barrier()
        if hasattr(graph, 'close'):
            graph.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
