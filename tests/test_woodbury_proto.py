"""The numpy statement of the covariance-form factor update (tests/woodbury_proto.py: one inverse per variable, 2x2 algebra per factor,
relinearisation as a rank-2 downdate) against the C oracle's dense reference maths, sweep by sweep through ba.py's schedule -- the
algebra the HIP kernels implement, readable in forty lines of numpy."""
import os

import numpy as np

from conftest import DATA, rel_err_rows
from woodbury_proto import WoodburyBA


def test_covariance_form_equals_the_dense_reference_maths(oracle_mod):
    from gbp_amd.balio import read_bal
    prob = read_bal(os.path.join(DATA, 'fr1desk_vsmall.txt'), native=False)
    o = oracle_mod.OracleBA.from_problem(prob)
    w = WoodburyBA(prob)
    for g in (o, w):
        g.generate_priors_var(50.0)
        g.update_all_beliefs()
    order = np.argsort(prob.cam_idx, kind='stable')
    n_relin, worst = 0, 0.0
    for it in range(26):
        if it in (3, 8):
            o.set_iters_since_relin(1)
            w.iters[:] = 1
        o.synchronous_iteration(robustify=True, local_relin=True)
        n_relin += w.synchronous_iteration()
        worst = max(worst, max(rel_err_rows(a, b) for a, b in zip((w.cam_eta, w.cam_lam, w.lmk_eta, w.lmk_lam), o.beliefs())))
        assert np.array_equal(w.iters[order], o.relin_state()['iters_since_relin'])
    assert n_relin >= 2 * prob.n_factors - 50          # sweeps 16 and 25 relinearise (nearly) everybody: the downdate path ran
    assert worst < 1e-6, worst
