#!/usr/bin/env python3
"""Throughput of the linear pairwise GBP engine (include/gbp_lin.h) on a large synthetic pose graph (secondary path;
the headline benchmark is bench.py).  Variables on a ring, each joined to its next `k` neighbours by a
linear_displacement factor (gbp/factors/linear_displacement.py:8-14).  Prints one JSON line.

Algorithmic bytes per sweep (fp64, packed symmetric, P = d(d+1)/2): per factor read Lambda_f d(2d+1) + eta_f 2d +
two belief records 2(d+P) + two old messages 2(d+P), write two messages 2(d+P), and the belief stage reads them again
2(d+P); per variable prior d+P read, d+P+d written."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gbp_amd.linear import LinearEngine

ap = argparse.ArgumentParser()
ap.add_argument('--vars', type=int, default=200_000)
ap.add_argument('--dofs', type=int, default=3)
ap.add_argument('--k', type=int, default=5)
ap.add_argument('--steps', type=int, default=100)
ap.add_argument('--warmup', type=int, default=10)
ap.add_argument('--no-cpu-baseline', action='store_true')
a = ap.parse_args()
rs = np.random.RandomState(0)
N, D, k = a.vars, a.dofs, a.k
mu0 = rs.rand(N, D) * 10
va = np.repeat(np.arange(N), k)
vb = (va + np.tile(np.arange(1, k + 1), N)) % N
F = va.shape[0]
z = mu0[vb] - mu0[va] + rs.normal(0, 1.0, (F, D))
J = np.hstack([-np.eye(D), np.eye(D)])
fe = z @ J                                  # J^T z per factor, sigma = 1
fl = np.broadcast_to(J.T @ J, (F, 2 * D, 2 * D))
fc = 0.5 * np.einsum('fd,fd->f', z, z)
pl = np.broadcast_to(np.eye(D) / 3.0, (N, D, D))
e = LinearEngine(va, vb, fe, np.ascontiguousarray(fl), mu0 / 3.0, np.ascontiguousarray(pl), factor_const=fc)
e.update_all_beliefs()
e.iterate(a.warmup); e.sync()
t0 = time.perf_counter(); e.iterate(a.steps); e.sync(); dt = time.perf_counter() - t0
P = D * (D + 1) // 2
bytes_sweep = 8 * (F * (D * (2 * D + 1) + 2 * D + 6 * (D + P) + 2 * (D + P)) + N * (2 * (D + P) + D))
cpu = None
if not a.no_cpu_baseline:
    # bounded sample of the same workload for the numpy oracle (dense per-factor loops, one thread): the first 400 variables
    # of the ring with their factors, 3 sweeps; reported per factor-sweep and scaled to this graph
    from oracle.linear_oracle import LinearOracle
    n_s = 400
    sel = (va < n_s - k)
    o = LinearOracle(va[sel], vb[sel], fe[sel], np.ascontiguousarray(fl[sel]), (mu0 / 3.0)[:n_s], np.ascontiguousarray(pl[:n_s]), factor_const=fc[sel])
    o.update_all_beliefs(); o.synchronous_iteration()
    tc = time.perf_counter(); o.iterate(3); tc = (time.perf_counter() - tc) / 3
    us_per_factor = 1e6 * tc / int(sel.sum())
    cpu = {"value": 1.0 / (us_per_factor * 1e-6 * F), "unit": "iter/s", "cores": 1, "kind": "port",
           "sample": f"numpy oracle (oracle/linear_oracle.py) on the first {n_s} variables / {int(sel.sum())} factors of the same ring, "
                     f"3 sweeps, {us_per_factor:.1f} us per factor-sweep, scaled to {F} factors"}
print(json.dumps({"metric": "linear GBP sweeps/s", "value": a.steps / dt, "unit": "iter/s", "ms_per_step": 1e3 * dt / a.steps,
                  "config": {"workload": f"ring pose graph {N} vars x {D} dofs, {F} linear_displacement factors"},
                  "dtype": "f64", "roofline": {"bound": "hbm", "achieved": bytes_sweep * a.steps / dt / 1e9, "peak": 8000.0,
                                               "unit": "GB/s", "frac": bytes_sweep * a.steps / dt / 1e9 / 8000.0, "traffic": None},
                  "energy_after": e.energy(), "cpu_baseline": cpu}))
