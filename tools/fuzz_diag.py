#!/usr/bin/env python3
"""Diagnose one seed of tests/test_fuzz_gpu.py: per sweep, the belief gap of both GPU sweeps against the C oracle and against
each other, next to the conditioning of the cavities (a gap that follows the condition number is two correct eliminations
disagreeing by cond x eps; a gap between the two GPU sweeps would be a bug).   PYTHONPATH=. python tools/fuzz_diag.py SEED..."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
from conftest import rel_err_rows
import test_fuzz_gpu as tf
from gbp_amd.engine import BAEngine
from gbp_amd.balio import reference_factor_order
from oracle import oracle as om
from oracle.numpy_ba import NumpyBA
om.build()
for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(seed)
    p = tf.random_problem(seed)
    loss = [None, 'huber', 'constant'][seed % 3]
    cfg = dict(loss=loss, Nstds=float(rng.uniform(1.0, 3.0)), beta=float(rng.choice([0.005, 0.01, 0.05])),
               num_undamped_iters=int(rng.choice([1, 2, 6])), min_linear_iters=int(rng.choice([2, 4, 8])),
               eta_damping=float(rng.choice([0.3, 0.4, 0.7])), gauss_noise_std=float(rng.uniform(1.5, 3.0)))
    flags = [(bool(rng.integers(0, 2)), bool(rng.random() < 0.8)) for _ in range(8)]
    print(f"seed {seed}: C={p.n_cams} L={p.n_lmks} F={p.n_factors} {cfg}")
    o = om.OracleBA.from_problem(p, threads=4, **cfg)
    ef, eg = BAEngine.from_problem(p, fused=True, **cfg), BAEngine.from_problem(p, fused=False, **cfg)
    nb = NumpyBA(p, **cfg)                                    # third opinion: object per factor, np.linalg.inv like the reference
    for g in (o, ef, eg, nb):
        g.generate_priors_var(30.0); g.update_all_beliefs()
    are0 = o.are()
    order = reference_factor_order(p.cam_idx)
    cam, lmk = p.cam_idx[order], p.lmk_idx[order]
    for i, (rob, rel) in enumerate(flags):
        for g in (o, ef, eg, nb):
            g.synchronous_iteration(robustify=rob, local_relin=rel)
        ob = o.beliefs()
        gf = max(rel_err_rows(a, b) for a, b in zip(ef.beliefs(), ob))
        gg = max(rel_err_rows(a, b) for a, b in zip(eg.beliefs(), ob))
        fg = max(rel_err_rows(a, b) for a, b in zip(ef.beliefs(), eg.beliefs()))
        no = max(rel_err_rows(a, b) for a, b in zip(nb.beliefs(), ob))
        nf = max(rel_err_rows(a, b) for a, b in zip(nb.beliefs(), ef.beliefs()))
        _, cl, _, ll = ob
        _, mcl, _, mll = o.messages()
        ec = np.linalg.eigvalsh(cl[cam] - mcl); el = np.linalg.eigvalsh(ll[lmk] - mll)
        bc = np.linalg.eigvalsh(cl); bl = np.linalg.eigvalsh(ll)
        print(f"  sweep {i} rob={rob} rel={rel}: fused-oracle {gf:.2e} general-oracle {gg:.2e} fused-general {fg:.2e} numpy-oracle {no:.2e} numpy-fused {nf:.2e} | cavity cond cam {np.max(ec[:, -1] / ec[:, 0]):.2e} "
              f"lmk {np.max(el[:, -1] / el[:, 0]):.2e} min eig {ec.min():.2e} {el.min():.2e} | belief cond cam {np.max(bc[:, -1] / bc[:, 0]):.2e} lmk {np.max(bl[:, -1] / bl[:, 0]):.2e} "
              f"| healthy {tf.healthy(o, p, are0)} are {o.are():.3g} energy o {o.energy():.9g} f {ef.energy():.9g}")
