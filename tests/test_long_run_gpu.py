"""GPU parity at the reference's own run length: ba.py defaults to --n_iters 200 (ba.py:13) and runs all of them (ba.py:84-105).

Fixture G14 (tests/golden/make_golden.py: g14) is the REFERENCE itself through those 200 sweeps on fr1desk_vsmall and fr1desk_small with
ba.py's default flags and on fr1desk_vsmall with --loss huber: ARE / energy / number of freshly relinearised factors before every
sweep, all beliefs and every factor's iters_since_relin after sweeps 30, 40, 50, 75, 100, 125, 150, 175, 200.  From sweep ~60 on some
factors relinearise in EVERY sweep (a few dozen to a few hundred of 1 801 / 3 917), so "the same factors relinearise in the same sweep"
is asserted 140 times in a row: had the engine's trajectory parted from the reference's anywhere -- one factor crossing the beta
threshold one sweep early -- the counts and the per-factor ages would differ from there on.  They do not, on any of the three sweeps
the product can pick (fused, general, the automatic choice)."""
import os

import numpy as np
import pytest

from conftest import DATA, G15B_HOLD, G15B_NEAR, belief_gap, golden
from gbp_amd.balio import read_bal

pytestmark = pytest.mark.gpu

BELIEF_TOL = 1e-6           # (observed: ~1e-8; BASELINE north_star asks for 1e-4)


@pytest.mark.parametrize('fused', [True, False, None], ids=['fused', 'general', 'auto'])
@pytest.mark.parametrize('tag,loss', [('vsmall', None), ('small', None), ('vsmall_huber', 'huber'), ('desk', None)])
def test_g14_ba_default_length(oracle_mod, tag, loss, fused):
    from gbp_amd.engine import BAEngine
    g = golden(f'G14_200it_{tag}')
    p = read_bal(os.path.join(DATA, str(g['bal'])))
    e = BAEngine.from_problem(p, loss=loss, fused=fused)
    if fused is not None:
        assert e.info()['fused'] == fused
    e.generate_priors_var(50.0)
    e.update_all_beliefs()
    checkpoints = [int(c) for c in g['checkpoints']]
    relin, gaps, ages_off = [], {}, {}

    def grab(i, graph):                                     # top of loop index i = state after i sweeps (ba.py:96-99)
        relin.append(graph.count_relinearising())
        if i in checkpoints:
            gaps[i] = belief_gap(graph.beliefs(), g, f'it{i}_')
            ages_off[i] = int((graph.iters_since_relin() != g[f'it{i}_iters_since_relin']).sum())
        if i == 200 and loss:
            final.update(graph.relin_state())
    final = {}
    ares, energies = oracle_mod.replay_ba(e, 201, diagnostics=True, on_iter=grab)
    relin = np.array(relin[:200])
    first_fork = np.nonzero(relin != g['n_relin'])[0]
    assert first_fork.size == 0, f"relinearisation counts part from the reference's at sweep {first_fork[0]}: {relin[first_fork[0]]} vs {g['n_relin'][first_fork[0]]}"
    assert (g['n_relin'][60:] > 0).sum() > 100             # (the fixture really is the churning regime described above)
    assert np.allclose(ares[:200], g['are'], rtol=1e-6) and np.allclose(energies[:200], g['energy'], rtol=1e-5)
    assert ares[200] == pytest.approx(float(g['are_final']), rel=1e-6) and energies[200] == pytest.approx(float(g['energy_final']), rel=1e-5)
    assert sorted(gaps) == checkpoints
    assert all(v == 0 for v in ages_off.values()), ages_off
    assert max(gaps.values()) < BELIEF_TOL, gaps
    if loss:
        assert np.allclose(final['adaptive_var'], g['adaptive_var'], rtol=1e-8)
        assert np.array_equal(final['robust_flag'], g['robust_flag'])
    e.close()


# Fixture G15 runs in a regime that is ill-conditioned BY CONSTRUCTION (priors at 1 / 250 000 of the factors' information, landmarks seen
# twice): two float64 implementations of the reference's own formulas part there.  The C oracle -- the reference's dense arithmetic with
# another inverse routine -- is 1e-8 from the reference until the first relinearisation wave and 1e-6 ... 5e-5 after it (beliefs; ARE up
# to 3e-4 on single sweeps), with the SAME factors relinearising in every sweep.  The bound below is BASELINE's own 1e-4; the tight 1e-6
# is asserted up to the first wave.
G15_BELIEF_TOL, G15_ARE_TOL = 1e-4, 1e-3


@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'general'])
@pytest.mark.parametrize('tag', ['vsmall', 'small'])
def test_g15_float_implementation_through_relinearisation(oracle_mod, tag, fused):
    """ba.py --float_implementation (priors weakened to 1 / 250 000 of the factors' information) through three waves in which every factor
    relinearises: 40 sweeps, the reference's own run (fixture G15).  This is where the covariance form could lose digits -- a landmark
    with two observations is held by two rank-2 messages and almost no prior, and the engine keeps its belief as mean | covariance and
    shows eta | Lambda as a view (Lambda = Sigma^-1) -- so the view itself is held against the reference here."""
    from gbp_amd.engine import BAEngine
    g = golden(f'G15_floatimpl_40it_{tag}')
    p = read_bal(os.path.join(DATA, str(g['bal'])))
    assert int(g['lmk_degree'].min()) == 2 and (g['n_relin'] == p.n_factors).sum() >= 3
    e = BAEngine.from_problem(p, fused=fused)
    e.generate_priors_var(50.0)
    e.update_all_beliefs()
    checkpoints = (12, 17, 26, 35, 40)
    relin, gaps = [], {}

    def grab(i, graph):
        relin.append(graph.count_relinearising())
        if i in checkpoints:
            gaps[i] = belief_gap(graph.beliefs(), g, f'it{i}_')
    ares, energies = oracle_mod.replay_ba(e, 41, diagnostics=True, on_iter=grab, float_impl=True)
    assert np.array_equal(np.array(relin[:40]), g['n_relin'])
    assert np.allclose(ares[:16], g['are'][:16], rtol=1e-6) and np.allclose(ares[:40], g['are'], rtol=G15_ARE_TOL)
    assert sorted(gaps) == list(checkpoints) and gaps[12] < 1e-6 and max(gaps.values()) < G15_BELIEF_TOL, gaps
    e.close()


@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'general'])
@pytest.mark.parametrize('tag', ['vsmall', 'small'])
def test_g15b_float_implementation_at_ba_default_length(oracle_mod, tag, fused):
    """`ba.py --float_implementation` at ba.py's default length, against the reference's own run as far as it gets (fixture G15b; conftest
    G15B_HOLD says what the run is: the REFERENCE diverges under this flag -- ARE 2.7 -> 20 558 px on fr1desk_small -- and dies of a
    singular matrix in sweep 143).  Asserted while the trajectory is still contracting: beliefs within BASELINE's 1e-4 at every checkpoint
    up to G15B_HOLD, every per-factor age equal there, relinearisation counts exact and the ARE within 1e-3 up to the hold sweep.  The gap
    at the later checkpoints is printed (and recorded by tests/tools/g15b_trace.py into profiles/), not bounded: the oracle leaves the
    reference at the same sweeps."""
    from gbp_amd.engine import BAEngine
    g = golden(f'G15b_floatimpl_200it_{tag}')
    p = read_bal(os.path.join(DATA, str(g['bal'])))
    e = BAEngine.from_problem(p, fused=fused)
    e.generate_priors_var(50.0)
    e.update_all_beliefs()
    hold_cp, hold_sweep = G15B_HOLD[tag]
    n_ref = len(g['are'])
    checkpoints = [int(c) for c in g['checkpoints'] if f'it{int(c)}_cam_eta' in g and int(c) <= 100]      # (beyond: tests/tools/g15b_trace.py)
    relin, gaps, ages = [], {}, {}

    def grab(i, graph):
        relin.append(graph.count_relinearising())
        if i in checkpoints:
            gaps[i] = belief_gap(graph.beliefs(), g, f'it{i}_')
            ages[i] = int((graph.iters_since_relin() != g[f'it{i}_iters_since_relin']).sum())
    ares, _ = oracle_mod.replay_ba(e, min(n_ref, max(checkpoints) + 1), diagnostics=True, on_iter=grab, float_impl=True)
    print(f"G15b {tag} {'fused' if fused else 'general'}: belief gap per checkpoint", {k: f'{v:.1e}' for k, v in gaps.items()})
    assert np.array_equal(np.array(relin[:hold_sweep]), g['n_relin'][:hold_sweep])
    assert np.allclose(ares[:hold_sweep], g['are'][:hold_sweep], rtol=1e-3)
    held = {k: v for k, v in gaps.items() if k <= hold_cp}
    assert max(held.values()) < 1e-4 and all(ages[k] == 0 for k in held), (gaps, ages)
    if tag in G15B_NEAR:
        k, bound = G15B_NEAR[tag]
        assert gaps[k] < bound, gaps
    e.close()
