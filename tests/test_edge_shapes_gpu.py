"""Deterministic graphs at the exact boundaries of the engine's plans (the fuzz only meets them by chance):
the last camera count whose table fits the LDS / the first that does not (fused <-> general sweep), a landmark that fills
a 64-slot tile exactly / one factor more (chunk tiles + k_lmk_belief_list), 24 / 25 landmarks in a tile."""
import numpy as np
import pytest

from conftest import rel_err_rows
from gbp_amd.synthetic import BAProblem, make_synthetic

pytestmark = pytest.mark.gpu

BELIEF_TOL = 1e-6


def run_pair(oracle_mod, prob, n_sweeps=14, **kw):
    from gbp_amd.engine import BAEngine
    o = oracle_mod.OracleBA.from_problem(prob, threads=8)
    e = BAEngine.from_problem(prob, **kw)
    for g in (o, e):
        g.generate_priors_var(50.0)
        g.update_all_beliefs()
        oracle_mod.replay_ba(g, n_sweeps)
    gap = max(rel_err_rows(a, b) for a, b in zip(e.beliefs(), o.beliefs()))
    assert np.array_equal(o.relin_state()['iters_since_relin'], e.relin_state()['iters_since_relin'])
    return gap, o, e


def test_camera_count_at_the_lds_boundary(oracle_mod, monkeypatch):
    from gbp_amd import _capi
    cmax = _capi.load().gbp_ba_fused_max_cams()
    monkeypatch.setenv('GBP_WINDOWS', '0')         # whole tables (with camera windows these 90 one-tile workgroups need some 60 rows each: last block)
    assert 256 <= cmax <= 758                       # 160 KB / 216 B per camera, minus the per-wave scratch
    # one table in LDS (fused sweep) / one camera more: the general sweep, the same loop with camera-major staging.  (5 400 factors on
    # ~590 cameras are sparse: left to itself the library would run the staged sweep at every one of these sizes; fused=True asks for
    # the fused sweep wherever its table fits, and above that it cannot be had.)
    for C, fused_path in ((cmax, 1), (cmax + 1, 0), (2 * cmax, 0)):
        prob = make_synthetic(n_cams=C, n_lmks=900, obs_per_lmk=6, seed=21)
        gap, o, e = run_pair(oracle_mod, prob, n_sweeps=10, fused=True)
        assert e.info()['cam_groups'] == fused_path, (C, e.info())
        assert e.plan_info()['staged_by_sparseness'] is False
        assert gap < BELIEF_TOL, (C, gap)
        for a, b in zip(e.messages(), o.messages()):
            assert rel_err_rows(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)) < 1e-5
    monkeypatch.delenv('GBP_WINDOWS')
    prob = make_synthetic(n_cams=2 * cmax, n_lmks=900, obs_per_lmk=6, seed=21)
    gap, o, e = run_pair(oracle_mod, prob, n_sweeps=10, fused=True)
    pi = e.plan_info()
    assert pi['fused'] and 0 < pi['max_window'] <= 64 and gap < BELIEF_TOL, (pi, gap)      # random cameras, but one tile per workgroup: sets of at most 64


def test_sparse_graphs_take_the_staged_sweep(oracle_mod, monkeypatch):
    """Few factors per (workgroup, camera).  Left alone the library runs the fused sweep with camera windows (each workgroup's table holds
    the few dozen cameras of its own tile).  With whole tables only (GBP_WINDOWS=0) it runs the staged sweep although the camera table
    would fit the LDS (one 128-byte row per factor instead of one 224-byte table row per camera and workgroup); a dense graph keeps
    the fused sweep."""
    monkeypatch.delenv('GBP_STAGED_BELOW', raising=False)
    sparse = make_synthetic(n_cams=400, n_lmks=1500, obs_per_lmk=6, seed=4)          # 9 000 factors, 150 tiles x 400 cameras
    monkeypatch.delenv('GBP_WINDOWS', raising=False)
    gap, o, e = run_pair(oracle_mod, sparse, n_sweeps=12)
    pi = e.plan_info()
    assert pi['fused'] and 0 < pi['max_window'] <= 64 and pi['table_rows'] <= sparse.n_factors and not pi['staged_by_sparseness'] and gap < BELIEF_TOL, (pi, gap)
    monkeypatch.setenv('GBP_WINDOWS', '0')
    gap, o, e = run_pair(oracle_mod, sparse, n_sweeps=12)
    assert e.info()['cam_groups'] == 0 and gap < BELIEF_TOL, (e.info(), gap)
    assert e.plan_info()['staged_by_sparseness'] is True
    gap, o, e = run_pair(oracle_mod, sparse, n_sweeps=12, fused=True)                # ... unless the caller insists (GBP_FLAG_FORCE_FUSED)
    assert e.info()['cam_groups'] == 1 and e.plan_info()['staged_by_sparseness'] is False and gap < BELIEF_TOL, (e.info(), gap)
    dense = make_synthetic(n_cams=12, n_lmks=1500, obs_per_lmk=6, seed=4)
    gap, o, e = run_pair(oracle_mod, dense, n_sweeps=12)
    assert e.info()['cam_groups'] == 1 and gap < BELIEF_TOL, (e.info(), gap)


@pytest.mark.parametrize('n_cams,window,obs,loss,single,closures', [(2000, 12, 6, None, None, 0.0), (2000, 12, 6, 'huber', '0', 0.0), (3000, 40, 10, 'constant', None, 0.0),
                                                                   (1500, 64, 40, None, None, 0.0), (300, 10, 5, None, None, 0.0),
                                                                   (2000, 12, 6, None, None, 0.03), (5000, 30, 10, 'huber', None, 0.05)])
def test_camera_windows_of_a_sequence(oracle_mod, monkeypatch, n_cams, window, obs, loss, single, closures):
    """A sequence: every landmark is seen from `obs` of `window` consecutive cameras, landmarks numbered along the trajectory.  The cameras
    of a workgroup's tiles form a short interval, so the fused sweep runs with per-workgroup camera WINDOWS -- with thousands of cameras,
    far beyond what one LDS table holds -- and gives the oracle's beliefs, messages and relinearisation ages (both accumulation variants,
    the robust losses, the dense packing at 40 factors per landmark, and a graph whose whole table WOULD fit but whose windows are
    much smaller).  closures: that share of the landmarks is seen from anywhere along the trajectory -- a workgroup's cameras then span
    the whole range, its SET is still a few dozen (the table rows go through a map over the interval)."""
    from gbp_amd import _capi
    cmax = _capi.load().gbp_ba_fused_max_cams()
    monkeypatch.delenv('GBP_WINDOWS', raising=False)
    if single is None:
        monkeypatch.delenv('GBP_ACC_SINGLE', raising=False)
    else:
        monkeypatch.setenv('GBP_ACC_SINGLE', single)
    n_lmks = 36_000 // obs
    prob = make_synthetic(n_cams=n_cams, n_lmks=n_lmks, obs_per_lmk=obs, seed=5, window=window, closures=closures)
    kw = dict(loss=loss, Nstds=2.0) if loss else {}
    from gbp_amd.engine import BAEngine
    o = oracle_mod.OracleBA.from_problem(prob, threads=8, **kw)
    e = BAEngine.from_problem(prob, **kw)
    pi = e.plan_info()
    assert pi['fused'] and not pi['staged_by_sparseness'], pi
    assert 0 < pi['max_window'] <= min(cmax, 3 * window + n_cams // 64 + int(closures * 36_000 / 128)), pi
    assert pi['table_rows'] < pi['n_blocks'] * n_cams // 2, pi
    assert pi['single'] == (single != '0'), pi
    if obs == 40:
        assert pi['pack_mode'] == 2, pi
    for g in (o, e):
        g.generate_priors_var(50.0)
        g.update_all_beliefs()
        oracle_mod.replay_ba(g, 14)
    assert np.array_equal(o.relin_state()['iters_since_relin'], e.relin_state()['iters_since_relin'])
    gap = max(rel_err_rows(a, b) for a, b in zip(e.beliefs(), o.beliefs()))
    assert gap < BELIEF_TOL, gap
    for a, b in zip(e.messages(), o.messages()):
        assert rel_err_rows(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)) < 1e-5
    before = [a.copy() for a in e.beliefs()]
    e.update_all_beliefs()
    for a, b in zip(e.beliefs(), before):
        assert rel_err_rows(a, b) < 1e-9
    assert e.check_layout() == 0 and e.are() == pytest.approx(o.are(), rel=1e-6)
    e.close()


def test_camera_windows_switch_and_fallbacks(oracle_mod, monkeypatch):
    """GBP_WINDOWS=0: whole tables (the same beliefs to rounding: only the order of the per-camera sums over workgroups differs);
    a sequence whose windows do not fit the LDS runs the general sweep; cameras nobody observes (no window holds them: no table row)
    keep their priors."""
    from gbp_amd import _capi
    from gbp_amd.engine import BAEngine
    cmax = _capi.load().gbp_ba_fused_max_cams()
    prob = make_synthetic(n_cams=300, n_lmks=7000, obs_per_lmk=5, seed=9, window=10)
    out = {}
    for sw in ('1', '0'):
        monkeypatch.setenv('GBP_WINDOWS', sw)
        e = BAEngine.from_problem(prob, fused=None if sw == '1' else True)      # (whole tables: 35 000 factors on 256 x 300 table rows are sparse, the staged sweep would run)
        assert (e.plan_info()['max_window'] > 0) == (sw == '1') and e.plan_info()['fused'], e.plan_info()
        e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(12)
        out[sw] = e.beliefs()
        e.close()
    assert max(rel_err_rows(a, b) for a, b in zip(out['1'], out['0'])) < 1e-10
    # the reduce behind the windows: one wave per camera (a handful of rows each), or -- a camera with more than 64 rows -- the tree
    monkeypatch.setenv('GBP_WINDOWS', '1')
    monkeypatch.setenv('GBP_ROWS_WAVE_MAX', '0')
    e = BAEngine.from_problem(prob)
    e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(12)
    assert e.plan_info()['max_window'] > 0 and max(rel_err_rows(a, b) for a, b in zip(out['1'], e.beliefs())) < 1e-10
    e.close()
    monkeypatch.delenv('GBP_ROWS_WAVE_MAX')
    monkeypatch.delenv('GBP_WINDOWS')
    wide = make_synthetic(n_cams=2 * cmax, n_lmks=40_000, obs_per_lmk=6, seed=9, window=cmax + 200)      # (940 factors per workgroup on ~790 cameras)
    gap, o, e = run_pair(oracle_mod, wide, n_sweeps=8)
    assert not e.plan_info()['fused'] and e.plan_info()['max_window'] == 0 and gap < BELIEF_TOL, (e.plan_info(), gap)
    e.close()
    # cameras 0..49 and the last 50 of 2100 have no factor
    seq = make_synthetic(n_cams=2000, n_lmks=6000, obs_per_lmk=6, seed=2, window=12)
    prob = BAProblem(K=seq.K, cam_means=np.concatenate([seq.cam_means[:50], seq.cam_means, seq.cam_means[-50:]]), lmk_means=seq.lmk_means,
                     meas=seq.meas, cam_idx=(seq.cam_idx + 50).astype(np.int32), lmk_idx=seq.lmk_idx)
    e = BAEngine.from_problem(prob)
    o = oracle_mod.OracleBA.from_problem(prob, threads=8)
    cov = [np.eye(6) * 1e-2] * prob.n_cams + [np.eye(3) * 1e-1] * prob.n_lmks
    e.set_priors_var(cov); o.set_priors_var(cov[:prob.n_cams], cov[prob.n_cams:])
    for g in (e, o):
        g.update_all_beliefs()
        g.iterate(8)
    assert e.plan_info()['max_window'] > 0, e.plan_info()
    assert max(rel_err_rows(a, b) for a, b in zip(e.beliefs(), o.beliefs())) < BELIEF_TOL
    e.close()


def test_handles_with_different_tables_stay_launchable(oracle_mod, monkeypatch):
    """The dynamic-LDS attribute of the sweep kernels belongs to the function, not to a handle: a handle with a 500-camera table (145 KB of
    LDS) keeps sweeping after handles with a 12-camera table and with camera windows were created in the same process, and all three
    give the oracle's beliefs."""
    from gbp_amd.engine import BAEngine
    monkeypatch.delenv('GBP_WINDOWS', raising=False)
    probs = [make_synthetic(n_cams=500, n_lmks=20_000, obs_per_lmk=10, seed=1), make_synthetic(n_cams=12, n_lmks=2000, obs_per_lmk=6, seed=2),
             make_synthetic(n_cams=2000, n_lmks=6000, obs_per_lmk=6, seed=3, window=12)]
    engines = []
    for p in probs:                                  # (each handle sweeps once before the next one is created)
        e = BAEngine.from_problem(p)
        e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(2); e.sync()
        engines.append(e)
    assert [e.plan_info()['max_window'] > 0 for e in engines] == [False, False, True]
    for e in engines:
        e.iterate(4)
    for p, e in zip(probs, engines):
        o = oracle_mod.OracleBA.from_problem(p, threads=8)
        o.generate_priors_var(50.0); o.update_all_beliefs(); o.iterate(6)
        assert max(rel_err_rows(a, b) for a, b in zip(e.beliefs(), o.beliefs())) < BELIEF_TOL
        e.close()


def test_tiles_past_the_memory_side_cache_change_nothing(monkeypatch, oracle_mod):
    """Graphs beyond the 256 MiB memory-side cache run the pinned variant of the fused sweep (FusedArgs::pin: the first tiles of a
    workgroup's walk keep using the cache, the rest stream past it with nontemporal loads and stores).  Cache hints only: forced on a
    small graph -- every tile streamed, half of them, none -- the beliefs must be BITWISE the same whatever the share.  Against the
    plain kernel they differ in the last bits since round 6 -- the pinned variant walks the tiles strided (workgroup b: tiles b, b + n,
    ...), so a camera's messages reach the workgroup tables in another order -- and both are held against the oracle."""
    from gbp_amd.engine import BAEngine
    prob = make_synthetic(n_cams=40, n_lmks=60_000, obs_per_lmk=6, seed=8)          # 360k factors: ~24 tiles per workgroup
    out = {}
    for keep in (None, '0', '60', '100000'):
        if keep is None:
            monkeypatch.delenv('GBP_FUSED_PIN_MIB', raising=False)
        else:
            monkeypatch.setenv('GBP_FUSED_PIN_MIB', keep)
        e = BAEngine.from_problem(prob)
        assert e.info()['cam_groups'] == 1
        assert (e.plan_info()['pinned_tiles'] >= 0) == (keep is not None)
        e.generate_priors_var(50.0)
        e.update_all_beliefs()
        e.set_iters_since_relin(8)                      # so that sweeps relinearise (x0 stores) as well
        e.iterate(12)
        out[keep] = [a.copy() for a in e.beliefs()] + [e.relin_state()['iters_since_relin'].copy()]
        e.close()
    for keep in ('60', '100000'):
        for a, b in zip(out[keep], out['0']):
            assert np.array_equal(a, b), keep
    o = oracle_mod.OracleBA.from_problem(prob, threads=8)
    o.generate_priors_var(50.0); o.update_all_beliefs(); o.set_iters_since_relin(8); o.iterate(12)
    for keep in (None, '0'):
        assert max(rel_err_rows(a, b) for a, b in zip(out[keep][:4], o.beliefs())) < BELIEF_TOL, keep
        assert np.array_equal(out[keep][4], o.relin_state()['iters_since_relin']), keep
    assert max(rel_err_rows(a, b) for a, b in zip(out['0'][:4], out[None][:4])) < 1e-9


def test_same_camera_lanes_in_one_atomic_instruction(monkeypatch):
    """Factors of a tile that hit the same camera must add to its LDS row in rank (= lane) order.  With many of them per tile (few
    cameras) the sweep issues ONE ds_add_f64 per entry for all lanes and relies on the LDS atomic unit applying same-address lanes in
    ascending lane order; with few it runs one round per rank.  Both must give the same bits -- this test is what pins that hardware
    behaviour: rounds (GBP_ACC_SINGLE=0), one instruction (=1) and the default (by the number of cameras) on the reference's own files
    and on a 20-camera synthetic graph."""
    import os
    from conftest import DATA
    from gbp_amd.balio import read_bal
    from gbp_amd.engine import BAEngine
    probs = [read_bal(os.path.join(DATA, 'fr1desk_small.txt')), read_bal(os.path.join(DATA, 'fr1desk.txt')),
             make_synthetic(n_cams=20, n_lmks=6000, obs_per_lmk=8, seed=3)]
    for prob in probs:
        out = {}
        for mode in ('0', '1', None):
            if mode is None:
                monkeypatch.delenv('GBP_ACC_SINGLE', raising=False)
            else:
                monkeypatch.setenv('GBP_ACC_SINGLE', mode)
            e = BAEngine.from_problem(prob)
            assert e.info()['cam_groups'] == 1
            e.generate_priors_var(50.0)
            e.update_all_beliefs()
            e.set_iters_since_relin(8)
            e.iterate(16)
            out[mode] = [a.copy() for a in e.beliefs()]
            e.close()
        for mode in ('1', None):
            assert all(np.array_equal(a, b) for a, b in zip(out[mode], out['0'])), mode


def test_single_variant_is_probed_on_the_device_and_falls_back(monkeypatch):
    """The plan of a graph that selects the one-instruction accumulation first verifies, on the device it will run on, the lane order
    that variant relies on (k_single_probe: seven address patterns, bitwise against lane-by-lane additions).  On this part the probe
    passes; told that it failed (GBP_SINGLE_PROBE_FAIL = a failing-pattern mask: test switch) the plan runs the rounds variant --
    same bits -- and gbp_ba_plan_info says so.  Graphs that never wanted the variant (many cameras) are not probed."""
    import os
    from conftest import DATA
    from gbp_amd.balio import read_bal
    from gbp_amd.engine import BAEngine
    prob = read_bal(os.path.join(DATA, 'fr1desk_small.txt'))
    out = {}
    for fail in (None, '5'):
        if fail is None:
            monkeypatch.delenv('GBP_SINGLE_PROBE_FAIL', raising=False)
        else:
            monkeypatch.setenv('GBP_SINGLE_PROBE_FAIL', fail)
        e = BAEngine.from_problem(prob, fused=True)
        pi = e.plan_info()
        assert pi['fused'] and pi['single'] == (fail is None) and pi['single_probe'] == (1 if fail is None else 0), pi
        e.generate_priors_var(50.0)
        e.update_all_beliefs()
        e.set_iters_since_relin(8)
        e.iterate(16)
        out[fail] = [a.copy() for a in e.beliefs()]
        e.close()
    assert all(np.array_equal(a, b) for a, b in zip(out['5'], out[None]))
    monkeypatch.delenv('GBP_SINGLE_PROBE_FAIL', raising=False)
    e = BAEngine.from_problem(make_synthetic(n_cams=500, n_lmks=20_000, obs_per_lmk=10, seed=1), fused=True)
    pi = e.plan_info()
    assert pi['fused'] and not pi['single'] and pi['single_probe'] == -1 and pi['pinned_tiles'] == -1 and pi['n_blocks'] == 256 and pi['pack_mode'] == 0, pi
    e.close()


def with_landmarks(p, degrees, seed=5):
    """p plus one landmark per entry of `degrees`, seen by that many cameras (points near the origin are in front of
    and inside the image of every camera of the generator's shell)."""
    from gbp_amd.synthetic import rodrigues
    rng = np.random.default_rng(seed)
    R = rodrigues(p.cam_means[:, 3:6])
    lm, me, ci, li = [p.lmk_means], [p.meas], [p.cam_idx], [p.lmk_idx]
    for j, deg in enumerate(degrees):
        pt = rng.uniform(-0.3, 0.3, 3)
        cams = np.sort(rng.permutation(p.n_cams)[:deg]).astype(np.int32)
        pc = np.einsum('cij,j->ci', R[cams], pt) + p.cam_means[cams, 0:3]
        assert (pc[:, 2] > 0.5).all()
        uv = np.stack([p.K[0] * pc[:, 0] / pc[:, 2] + p.K[2], p.K[1] * pc[:, 1] / pc[:, 2] + p.K[3]], axis=1)
        lm.append((pt + rng.normal(scale=0.05, size=3))[None]); me.append(uv + rng.normal(scale=1.0, size=uv.shape))
        ci.append(cams); li.append(np.full(deg, p.n_lmks + j, np.int32))
    # keep the file camera-major like the generator's output
    cam_idx, lmk_idx, meas = np.concatenate(ci), np.concatenate(li), np.concatenate(me)
    order = np.argsort(cam_idx, kind='stable')
    return BAProblem(K=p.K, cam_means=p.cam_means, lmk_means=np.concatenate(lm), meas=meas[order],
                     cam_idx=cam_idx[order].astype(np.int32), lmk_idx=lmk_idx[order].astype(np.int32))


@pytest.mark.parametrize('fused', [True, False])
def test_landmark_degree_64_and_65(oracle_mod, fused):
    """63 / 64 fill one tile (whole landmark owned by the tile), 65 / 128 / 129 become chunk tiles: each adds up its piece of the
    landmark's messages (Params::parts) and k_lmk_finish_parts forms the belief."""
    base = make_synthetic(n_cams=140, n_lmks=60, obs_per_lmk=5, seed=31)
    prob = with_landmarks(base, [64, 65, 63, 128, 129, 2])
    gap, o, e = run_pair(oracle_mod, prob, fused=fused)
    assert gap < BELIEF_TOL, gap
    for a, b in zip(e.messages(), o.messages()):
        assert rel_err_rows(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)) < 1e-5


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('obs,n_lmks,pack', [(40, 120, None), (100, 50, None), (23, 200, None), (10, 300, 'dense'), (3, 700, 'dense'), (7, 301, 'dense'),
                                             (40, 120, 'whole')])
def test_dense_packing_landmarks_span_tiles(oracle_mod, monkeypatch, fused, obs, n_lmks, pack):
    """Dense packing (tile t = factors [64 t, 64 t + 64) of the landmark-major list; gbp_build.hpp): picked by the library when whole
    landmarks would leave more than 15 % of the slots empty (40 or 23 factors per landmark: one or two per tile; 100: chunk tiles of
    64 + 36), forced here on shapes it would not pick it for (GBP_PACK=dense: 10, 3, 7 factors per landmark -- parts of 1..9 factors at
    every tile boundary, 21 landmarks in a tile) and forbidden on one it would (GBP_PACK=whole).  Same beliefs, messages and
    relinearisation ages as the oracle either way, on the fused and on the general sweep."""
    if pack:
        monkeypatch.setenv('GBP_PACK', pack)
    else:
        monkeypatch.delenv('GBP_PACK', raising=False)
    prob = make_synthetic(n_cams=130, n_lmks=n_lmks, obs_per_lmk=obs, seed=61)
    gap, o, e = run_pair(oracle_mod, prob, fused=fused)
    pi = e.plan_info()
    want_dense = pack == 'dense' or (pack is None)
    assert pi['pack_mode'] == (2 if want_dense else 0), pi
    assert pi['n_tiles'] == ((prob.n_factors + 63) // 64 if want_dense else n_lmks * ((obs + 63) // 64)), pi
    assert gap < BELIEF_TOL, gap
    for a, b in zip(e.messages(), o.messages()):
        assert rel_err_rows(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)) < 1e-5
    # the other entry points that walk the layout: update_all_beliefs from the stored messages, the residual, the layout check
    before = [a.copy() for a in e.beliefs()]
    e.update_all_beliefs()
    for a, b in zip(e.beliefs(), before):
        assert rel_err_rows(a, b) < 1e-9
    assert e.check_layout() == 0 and e.are() == pytest.approx(o.are(), rel=1e-6)


def test_dense_packing_needs_three_factors_per_landmark(monkeypatch):
    """One landmark of two factors: a 64-factor window could then touch more than 24 landmarks, so the dense packing is not offered
    (not even on request) and the whole-landmark packing runs."""
    from gbp_amd.engine import BAEngine
    monkeypatch.setenv('GBP_PACK', 'dense')
    prob = with_landmarks(make_synthetic(n_cams=130, n_lmks=100, obs_per_lmk=40, seed=7), [2])
    e = BAEngine.from_problem(prob)
    assert e.plan_info()['pack_mode'] == 0 and e.info()['n_tiles'] == 100      # (the two-factor landmark shares a tile with a 40-factor one)
    e.close()


def test_tile_led_by_landmarks_without_factors(oracle_mod):
    """Landmarks nobody observes take no slot but are owned by a tile (their belief is their prior).  Ten of them in front of landmarks
    WITH factors: the tile's belief phase adds up seven landmarks per pass and writes the sums of pass b into scratch rows 7 b .. 7 b + 6,
    which must not hold messages a later pass has yet to read -- "landmark l's factors sit in rows >= l" does not survive empty
    landmarks, so the packer starts a new tile where a run of them would pull a later landmark's rows forward (rounds 1-4 did not:
    such a tile summed sums)."""
    from gbp_amd.engine import BAEngine
    base = make_synthetic(n_cams=12, n_lmks=9, obs_per_lmk=5, seed=77)
    empty = 10
    prob = BAProblem(K=base.K, cam_means=base.cam_means,
                     lmk_means=np.concatenate([np.random.default_rng(1).uniform(-1, 1, (empty, 3)), base.lmk_means]), meas=base.meas,
                     cam_idx=base.cam_idx, lmk_idx=(base.lmk_idx + empty).astype(np.int32))
    for fused in (True, False):
        e = BAEngine.from_problem(prob, fused=fused)
        o = oracle_mod.OracleBA.from_problem(prob)
        cov = [np.eye(6) * 1e-2] * prob.n_cams + [np.eye(3) * 1e-1] * prob.n_lmks     # (generate_priors_var gives an unobserved landmark a zero prior)
        for g in (e, o):
            g.set_priors_var(cov) if g is e else g.set_priors_var(cov[:prob.n_cams], cov[prob.n_cams:])
            g.update_all_beliefs()
            g.iterate(6)
        assert e.info()['n_tiles'] == 2                      # (the ten empty landmarks | the nine with factors: landmark 10 would have been landmark 10 of its tile -- pass 1 -- with its factors in rows 0-4)
        gap = max(rel_err_rows(a, b) for a, b in zip(e.beliefs(), o.beliefs()))
        assert gap < BELIEF_TOL, (fused, gap)
        e.close()


@pytest.mark.parametrize('n_lmks,n_tiles', [(24, 1), (25, 2), (48, 2), (49, 3)])
def test_24_and_25_landmarks_per_tile(oracle_mod, n_lmks, n_tiles):
    prob = make_synthetic(n_cams=12, n_lmks=n_lmks, obs_per_lmk=2, seed=41)
    gap, o, e = run_pair(oracle_mod, prob)
    assert e.info()['n_tiles'] == n_tiles
    assert gap < BELIEF_TOL, gap


@pytest.mark.parametrize('obs,n_lmks,n_tiles', [(4, 16, 1), (4, 17, 2), (3, 21, 1), (3, 22, 2)])
def test_tile_filled_to_the_last_slot(oracle_mod, obs, n_lmks, n_tiles):
    """16 x 4 = 64 slots used exactly; 21 x 3 = 63 (the 22nd landmark does not fit the one free slot)."""
    prob = make_synthetic(n_cams=12, n_lmks=n_lmks, obs_per_lmk=obs, seed=43)
    gap, o, e = run_pair(oracle_mod, prob)
    assert e.info()['n_tiles'] == n_tiles
    assert gap < BELIEF_TOL, gap


def test_relinearisation_counters(oracle_mod):
    """Device-side counts of ba.py:96-99 against the per-factor state, and the saturating iters_since_relin."""
    from gbp_amd import _capi
    from gbp_amd.engine import BAEngine
    prob = make_synthetic(n_cams=10, n_lmks=300, obs_per_lmk=4, seed=51)
    e = BAEngine.from_problem(prob)
    e.generate_priors_var(50.0)
    e.update_all_beliefs()
    seen = []
    for i in range(20):
        e.iterate(1)
        st = e.relin_state()
        seen.append(int((st['iters_since_relin'] == 0).sum()))
        assert e.count_relinearising() == seen[-1]
    assert np.array_equal(e.relin_counts(20), seen)
    assert max(seen) > 0
    r = e.relin_state_range(100, 50)
    assert np.array_equal(r['iters_since_relin'], st['iters_since_relin'][100:150])
    assert np.array_equal(r['eta_damping'], st['eta_damping'][100:150])
    f = e.factors(100, 50, dense=False)
    full = e.factors(dense=False)
    assert np.array_equal(f['linpoint'], full['linpoint'][100:150]) and np.array_equal(f['z'], full['z'][100:150])
    # saturation: the counter stops at 2^19 - 1 instead of running into the sign bit of the state word
    top = (1 << 19) - 1
    e.set_iters_since_relin(top - 1)
    e.iterate(3, local_relin=True)
    st = e.relin_state()['iters_since_relin']
    assert ((st == top) | (st < 8)).all()           # saturated, or relinearised meanwhile
    with pytest.raises(_capi.GbpError):
        e.set_iters_since_relin(top + 1)
    with pytest.raises(_capi.GbpError):
        e.set_iters_since_relin(np.full(e.F, -1, np.int32))


def test_device_side_graph_build(oracle_mod):
    """gbp_ba_create orders, tiles and linearises on the device (gbp_build.hpp).  Shuffled observation rows take the stable
    radix sort by camera (every shipped file is already camera-major and skips it); the layout self-check kernel must find
    nothing; observations that already live on the GPU (GBP_FLAG_DEVICE_INPUT) give the bit-identical engine."""
    import torch
    from gbp_amd.engine import BAEngine
    rng = np.random.default_rng(3)
    base = with_landmarks(make_synthetic(n_cams=90, n_lmks=500, obs_per_lmk=7, seed=61), [70, 3, 64])
    perm = rng.permutation(base.n_factors)
    shuffled = BAProblem(K=base.K, cam_means=base.cam_means, lmk_means=base.lmk_means, meas=base.meas[perm],
                         cam_idx=base.cam_idx[perm], lmk_idx=base.lmk_idx[perm])
    out = []
    for prob in (base, shuffled):
        for fused in (True, False):
            gap, o, e = run_pair(oracle_mod, prob, n_sweeps=10, fused=fused)
            assert e.check_layout() == 0
            assert gap < BELIEF_TOL, gap
            f, g = e.factors(dense=False), o.factors()
            assert np.array_equal(f['cam'], g['cam']) and np.array_equal(f['lmk'], g['lmk']) and np.array_equal(f['z'], g['z'])
            out.append(e.beliefs())
    # the stable sort must reproduce the reference's order whatever the row order of the file: same engine, same bits per sweep type
    for a, b in zip(out[0], out[2]):
        assert rel_err_rows(a, b) < 1e-7           # different adj_factors order inside a camera only through file order
    dev = torch.device('cuda', 0)
    t = [torch.as_tensor(np.ascontiguousarray(x), device=dev) for x in (shuffled.cam_means, shuffled.lmk_means, shuffled.meas,
                                                                         shuffled.cam_idx.astype(np.int32), shuffled.lmk_idx.astype(np.int32))]
    torch.cuda.synchronize()
    e1 = BAEngine(shuffled.K, *[x.data_ptr() for x in t], device_pointers=(shuffled.n_cams, shuffled.n_lmks, shuffled.n_factors))
    e2 = BAEngine.from_problem(shuffled)
    for e in (e1, e2):
        e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(6)
    for a, b in zip(e1.beliefs(), e2.beliefs()):
        assert np.array_equal(a, b)
    assert e1.check_layout() == 0


def test_out_of_range_ids_are_refused_by_the_device_check():
    from gbp_amd import _capi
    from gbp_amd.engine import BAEngine
    p = make_synthetic(n_cams=6, n_lmks=40, obs_per_lmk=3, seed=1)
    for bad_cam, bad_lmk in ((6, 0), (-1, 0), (0, 40), (0, -2)):
        ci, li = p.cam_idx.copy(), p.lmk_idx.copy()
        ci[17] = bad_cam if bad_cam else ci[17]
        li[17] = bad_lmk if bad_lmk else li[17]
        with pytest.raises(_capi.GbpError):
            BAEngine(p.K, p.cam_means, p.lmk_means, p.meas, ci, li)


@pytest.mark.parametrize('num_undamped', [0, 1, 2])
def test_damping_in_the_relinearising_sweep(oracle_mod, num_undamped):
    """Message etas are stored as 2 coefficients in the rows of the Jacobian (gbp_math.hpp header), exact as long as a factor
    is never damped in the sweep it relinearises in.  num_undamped_iters = 0 is the one configuration where the reference does
    that (gbp.py:50-54): the engine then carries the out-of-span remainder densely and runs the general sweep."""
    from gbp_amd.engine import BAEngine
    prob = make_synthetic(n_cams=16, n_lmks=400, obs_per_lmk=5, seed=71)
    kw = dict(num_undamped_iters=num_undamped, min_linear_iters=3, eta_damping=0.4)
    o = oracle_mod.OracleBA.from_problem(prob, threads=8, **kw)
    e = BAEngine.from_problem(prob, **kw)
    assert e.info()['fused'] == (num_undamped > 0)
    relinearised = 0
    for g in (o, e):
        g.generate_priors_var(50.0)
        g.update_all_beliefs()
    for i in range(16):
        for g in (o, e):
            g.synchronous_iteration(robustify=True, local_relin=True)
        relinearised += int((e.relin_state()['iters_since_relin'] == 0).sum())
        gap = max(rel_err_rows(a, b) for a, b in zip(e.beliefs(), o.beliefs()))
        assert gap < BELIEF_TOL, (i, gap)
    assert relinearised > prob.n_factors                    # the case really occurred, more than once per factor
    for a, b in zip(e.messages(), o.messages()):
        assert rel_err_rows(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)) < 1e-5
    assert np.array_equal(o.relin_state()['eta_damping'], e.relin_state()['eta_damping'])


def test_degenerate_graphs(oracle_mod):
    """Shapes the device-side build must survive: no factors at all, landmarks nobody observes (in front, in the middle, at the
    end), a single factor, a camera without factors."""
    from gbp_amd.engine import BAEngine
    base = make_synthetic(n_cams=6, n_lmks=30, obs_per_lmk=3, seed=81)
    # no factors
    e = BAEngine(base.K, base.cam_means, base.lmk_means, np.zeros((0, 2)), np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert e.info()['n_tiles'] == 2 and e.check_layout() == 0           # 30 landmarks, 24 per tile, no slot used
    e.iterate(2)
    e.close()
    # unobserved landmarks: landmark ids stretched so that 0, 7, 8 and the last two have no factor
    remap = np.array([1, 2, 3, 4, 5, 6] + list(range(9, 33)))
    lm = np.concatenate([base.lmk_means[:1], base.lmk_means[:6], base.lmk_means[:2], base.lmk_means[6:], base.lmk_means[:2]])
    prob = BAProblem(K=base.K, cam_means=base.cam_means, lmk_means=lm, meas=base.meas, cam_idx=base.cam_idx,
                     lmk_idx=remap[base.lmk_idx].astype(np.int32))
    # a camera without factors: drop camera 5's observations
    keep = prob.cam_idx != 5
    prob = BAProblem(K=prob.K, cam_means=prob.cam_means, lmk_means=prob.lmk_means, meas=prob.meas[keep], cam_idx=prob.cam_idx[keep],
                     lmk_idx=prob.lmk_idx[keep])
    for fused in (True, False):
        o = oracle_mod.OracleBA.from_problem(prob)
        e = BAEngine.from_problem(prob, fused=fused)
        assert e.check_layout() == 0
        for g in (o, e):
            g.generate_priors_var(50.0)
        # variables without factors get a zero prior (max over no factors = 0, gbp_ba.py:27): their belief is singular in the
        # reference too, so the sweep is compared on the variables that have factors
        po, pe = o.priors(), e.priors()
        for a, b in zip(pe, po):
            assert np.allclose(a, b, rtol=1e-10, atol=0.0)
        assert not pe[1][5].any() and not pe[3][0].any() and not pe[3][-1].any()      # camera 5, first and last landmark: no factor
        e.close()
    # a single factor
    one = BAProblem(K=base.K, cam_means=base.cam_means[:1], lmk_means=base.lmk_means[:1], meas=base.meas[:1],
                    cam_idx=np.zeros(1, np.int32), lmk_idx=np.zeros(1, np.int32))
    gap, o, e = run_pair(oracle_mod, one, n_sweeps=6)
    assert gap < BELIEF_TOL and e.info()['n_tiles'] == 1


@pytest.mark.parametrize('fused', [True, False])
def test_device_resident_checkpoint(fused):
    """gbp_ba_snapshot_state / gbp_ba_restore_snapshot: the continuation from the restored state is bit-identical.  (General sweep: the
    staged camera rows keep the x0 half of a factor that does not relinearise from sweep to sweep -- a restore puts other linearisation
    points under them, and the next sweep must stage whole rows again.)"""
    from gbp_amd import _capi
    from gbp_amd.engine import BAEngine
    p = make_synthetic(n_cams=12, n_lmks=400, obs_per_lmk=5, seed=91)
    e = BAEngine.from_problem(p, fused=fused)
    with pytest.raises(_capi.GbpError):
        e.restore_snapshot()
    e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(7)
    e.snapshot_state()
    e.iterate(9)
    assert e.relin_counts(9).sum() > p.n_factors // 2      # linearisation points moved after the snapshot
    a = e.beliefs(); st_a = e.relin_state()['iters_since_relin']
    e.restore_snapshot()
    e.iterate(9)
    b = e.beliefs()
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and np.array_equal(st_a, e.relin_state()['iters_since_relin'])
    blob = e.save_state()                       # the host blob and the device slot are independent
    e.restore_snapshot(); e.iterate(2); e.load_state(blob)
    assert all(np.array_equal(x, y) for x, y in zip(a, e.beliefs()))


def test_rejected_state_blob_leaves_the_handle_alone():
    """gbp_ba_load_state validates version, sizes, graph hash and payload length BEFORE it touches the handle: a blob that
    claims a dense message remainder but is too short for one used to switch the handle to the general sweep and then fail."""
    from gbp_amd import _capi
    from gbp_amd.engine import BAEngine
    p = make_synthetic(n_cams=12, n_lmks=400, obs_per_lmk=5, seed=91)
    e = BAEngine.from_problem(p, fused=True)
    e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(3)
    blob = e.save_state()
    good = [a.copy() for a in e.beliefs()]
    bad = blob.copy()
    bad[20] |= 1                                  # StateHeader::reserved bit 0: "the payload ends with the dense remainder" -- it does not
    with pytest.raises(_capi.GbpError, match='truncated'):
        e.load_state(bad)
    assert e.info()['fused'] and e.plan_info()['fused']
    other = BAEngine.from_problem(make_synthetic(n_cams=12, n_lmks=400, obs_per_lmk=5, seed=92), fused=True)
    with pytest.raises(_capi.GbpError, match='different graph'):
        other.load_state(blob)
    assert other.info()['fused']
    e.iterate(2)
    e.load_state(blob)                            # the untouched handle still takes the good blob and continues from it
    assert all(np.array_equal(a, b) for a, b in zip(good, e.beliefs()))
