import os, sys, itertools
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gbp_amd.balio import read_bal
from gbp_amd import engine as eng
from oracle import oracle as oracle_mod
def rows(a, b):
    a = a.reshape(a.shape[0], -1); b = b.reshape(b.shape[0], -1)
    return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)
for name, beta, ml, nu, loss in itertools.product(['fr1desk_vsmall.txt', 'fr1desk_small.txt'], [0.01], [8, 6], [6, 3], [None, 'huber', 'constant']):
    p = read_bal('tests/golden/data/' + name)
    cfg = dict(loss=loss, Nstds=2.0, beta=beta, num_undamped_iters=nu, min_linear_iters=ml, eta_damping=0.4)
    o = oracle_mod.OracleBA.from_problem(p, threads=8, **cfg)
    e = eng.BAEngine.from_problem(p, fused=True, **cfg)
    for g in (o, e):
        g.generate_priors_var(50.0); g.update_all_beliefs()
    worst, nrel, asym = 0.0, 0, 0.0
    for rnd in range(22):
        if rnd % 3 == 2:
            for g in (o, e): g.synchronous_iteration(robustify=True, local_relin=True)
        else:
            for g in (o, e): g.robustify_all_factors()
            for g in (o, e): g.relinearise_factors()
            nrel += int((o.relin_state()['iters_since_relin'] == 0).sum())
            for g in (o, e): g.compute_all_messages(local_relin=True)
            for g in (o, e): g.update_all_beliefs()
        me, mo = e.messages(), o.messages()
        worst = max(worst, max(rows(a, b).max() for a, b in zip(me, mo)))
        asym = max(asym, np.abs(mo[1] - mo[1].transpose(0, 2, 1)).max() / np.abs(mo[1]).max())
    print(f"{name} beta {beta} min_linear {ml} undamped {nu} loss {loss}: worst msg gap {worst:.2e}, relinearised {nrel}, damped {(o.relin_state()['eta_damping'] > 0).sum()}, oracle asym {asym:.1e}", flush=True)
