"""The landmark-sharded driver with the REAL HIP engine at world sizes 2 and 3 on ONE GPU: every "rank" is a thread with
its own BAEngine (own stream) on device 0, and the collectives are a lock-step test double of torch.distributed (shared
slots + a barrier; gathers are concatenated in rank order exactly like all_gather_into_tensor).  This exercises what the
1-GPU box cannot do with RCCL: the in-library loop gbp_ba_iterate_sharded with n_ranks > 1 on real kernels (fused and
general sweeps) through a plugged-in exchange function, the Python-driven gbp_ba_shard_begin / _end path, the partition,
the MAX all-reduce of generate_priors_var and the globally normalised diagnostics."""
import os
import threading

import numpy as np
import pytest

from conftest import DATA, rel_err_rows
from gbp_amd.balio import read_bal
from gbp_amd.synthetic import make_synthetic

pytestmark = pytest.mark.gpu


class LockstepWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class LockstepDist:
    """The subset of torch.distributed that gbp_amd.sharded.ShardedBA uses, for threads of one process."""

    class ReduceOp:
        SUM, MAX = 'sum', 'max'

    def __init__(self, shared, rank):
        self.shared, self.rank = shared, rank

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.shared.world

    def _exchange(self, tensor):
        import torch
        torch.cuda.current_stream().synchronize()            # the producer kernels of this rank
        self.shared.slots[self.rank] = tensor
        self.shared.barrier.wait()
        parts = [t.clone() for t in self.shared.slots]
        torch.cuda.current_stream().synchronize()
        self.shared.barrier.wait()                           # everybody has copied before anyone overwrites
        return parts

    def device_exchange(self, rank, send_ptr, recv_ptr, count, stream_ptr):
        """gbp_exchange_fn of the in-library loop (gbp_ba_iterate_sharded): all-gather of `count` doubles per rank between
        raw device pointers, lock-step through host slots."""
        import ctypes as ct
        hip = ct.CDLL('libamdhip64.so')                      # the runtime the process already holds
        hip.hipStreamSynchronize.argtypes = [ct.c_void_p]
        hip.hipMemcpy.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int]
        assert hip.hipStreamSynchronize(ct.c_void_p(stream_ptr)) == 0          # this rank's producer kernels
        if count == 0:                                       # peer-store exchange: rendezvous only, the kernels moved the data
            self.shared.barrier.wait()
            return 0
        mine = np.empty(count)
        assert hip.hipMemcpy(mine.ctypes.data_as(ct.c_void_p), ct.c_void_p(send_ptr), count * 8, 2) == 0      # D2H
        self.shared.slots[rank] = mine
        self.shared.barrier.wait()
        allp = np.concatenate(self.shared.slots)
        assert hip.hipMemcpy(ct.c_void_p(recv_ptr), allp.ctypes.data_as(ct.c_void_p), allp.size * 8, 1) == 0   # H2D
        self.shared.barrier.wait()                           # everybody has copied before anyone overwrites
        return 0

    def all_gather_object(self, out_list, obj):
        self.shared.slots[self.rank] = obj
        self.shared.barrier.wait()
        out_list[:] = list(self.shared.slots)
        self.shared.barrier.wait()

    def barrier(self):
        self.shared.barrier.wait()

    def all_gather_into_tensor(self, out, inp):
        import torch
        out.copy_(torch.cat(self._exchange(inp)))

    def all_reduce(self, t, op=None):
        import torch
        parts = torch.stack(self._exchange(t))
        t.copy_(parts.max(dim=0).values if op == 'max' else parts.sum(dim=0))


def run_world(problem, world, fused, n_sweeps, oracle_mod, library_loop=True):
    from gbp_amd.sharded import ShardedBA
    shared = LockstepWorld(world)
    out, errors = [None] * world, []

    def rank_main(r):
        try:
            import torch
            torch.cuda.set_device(0)
            g = ShardedBA(problem, device=0, fused=fused, dist=LockstepDist(shared, r), library_loop=library_loop)
            assert g.library_loop == library_loop
            g.generate_priors_var(50.0)
            g.update_all_beliefs()
            ares, energies = oracle_mod.replay_ba(g, n_sweeps, diagnostics=True)
            ce, cl = g.camera_beliefs()
            rng, le, ll = g.local_landmark_beliefs()
            out[r] = dict(ce=ce, cl=cl, le=le, ll=ll, rng=rng, ares=ares, energies=energies, F=g.F, fused=g.info()['fused'], plan=g.engine.plan_info())
        except BaseException as e:                           # noqa: BLE001 -- surface it in the main thread
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    if errors:
        raise errors[0]
    return out


@pytest.mark.parametrize('world,fused,library_loop', [(2, True, True), (3, True, True), (2, False, True), (2, True, False)])
def test_sharded_engine_matches_single_engine(oracle_mod, world, fused, library_loop):
    from gbp_amd.engine import BAEngine
    p = read_bal(os.path.join(DATA, 'fr1desk_small.txt'))
    ref = BAEngine.from_problem(p, fused=fused)
    ref.generate_priors_var(50.0)
    ref.update_all_beliefs()
    ares, energies = oracle_mod.replay_ba(ref, 14, diagnostics=True)
    rce, rcl, rle, rll = ref.beliefs()
    ranks = run_world(p, world, fused, 14, oracle_mod, library_loop)
    assert sum(r['F'] for r in ranks) == p.n_factors
    lo = 0
    for r in ranks:
        assert r['fused'] == fused
        assert np.array_equal(r['ce'], ranks[0]['ce']) and np.array_equal(r['cl'], ranks[0]['cl'])   # identical on every rank
        assert rel_err_rows(r['ce'], rce) < 1e-6 and rel_err_rows(r['cl'], rcl) < 1e-6
        a, b = r['rng']
        assert a == lo
        lo = b
        assert rel_err_rows(r['le'], rle[a:b]) < 1e-6 and rel_err_rows(r['ll'], rll[a:b]) < 1e-6
        assert np.allclose(r['ares'], ares, rtol=1e-7) and np.allclose(r['energies'], energies, rtol=1e-7)
    assert lo == p.n_lmks


def test_sharded_engine_synthetic_two_ranks(oracle_mod):
    """A bigger, regular graph (8k factors per rank, several tiles per workgroup) through the fused shard kernels."""
    from gbp_amd.engine import BAEngine
    p = make_synthetic(n_cams=40, n_lmks=1600, obs_per_lmk=10, seed=4)
    ref = BAEngine.from_problem(p)
    ref.generate_priors_var(50.0); ref.update_all_beliefs(); ref.iterate(8)
    rce, rcl, rle, rll = ref.beliefs()
    shared_ranks = []
    from gbp_amd.sharded import ShardedBA
    shared = LockstepWorld(2)
    errors = []

    def rank_main(r):
        try:
            import torch
            torch.cuda.set_device(0)
            g = ShardedBA(p, device=0, dist=LockstepDist(shared, r))
            g.generate_priors_var(50.0); g.update_all_beliefs(); g.iterate(8)
            shared_ranks.append((r, g.camera_beliefs(), g.local_landmark_beliefs()))
        except BaseException as e:                           # noqa: BLE001
            errors.append(e); shared.barrier.abort()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    [t.start() for t in ts]; [t.join(300) for t in ts]
    if errors:
        raise errors[0]
    for r, (ce, cl), ((a, b), le, ll) in shared_ranks:
        assert rel_err_rows(ce, rce) < 1e-6 and rel_err_rows(cl, rcl) < 1e-6
        assert rel_err_rows(le, rle[a:b]) < 1e-6 and rel_err_rows(ll, rll[a:b]) < 1e-6


@pytest.mark.parametrize('windows', [False, True])
def test_sharded_engine_with_camera_groups_and_oversized_landmarks(oracle_mod, monkeypatch, windows):
    """700 cameras (more than ONE LDS table of the fused sweep holds) and landmarks seen by more than 64 cameras (chunk tiles whose
    beliefs run on the side stream beside the exchange) through the in-library loop at world size 2: on the general sweep
    (GBP_WINDOWS=0), and the way the library runs it left alone -- the fused sweep with camera windows (one or two tiles per workgroup:
    a few dozen of the 700 cameras each)."""
    from gbp_amd.engine import BAEngine
    from gbp_amd.synthetic import BAProblem
    if windows:
        monkeypatch.delenv('GBP_WINDOWS', raising=False)
    else:
        monkeypatch.setenv('GBP_WINDOWS', '0')
    big = make_synthetic(n_cams=700, n_lmks=3, obs_per_lmk=90, seed=8)
    q = make_synthetic(n_cams=700, n_lmks=1200, obs_per_lmk=6, seed=9)
    # the over-sized landmarks go to both ends so that each rank owns some
    lm = np.concatenate([big.lmk_means[:2], q.lmk_means, big.lmk_means[2:]])
    lidx = np.concatenate([big.lmk_idx[big.lmk_idx < 2], q.lmk_idx + 2, big.lmk_idx[big.lmk_idx == 2] - 2 + 2 + q.n_lmks])
    cidx = np.concatenate([big.cam_idx[big.lmk_idx < 2], q.cam_idx, big.cam_idx[big.lmk_idx == 2]])
    meas = np.concatenate([big.meas[big.lmk_idx < 2], q.meas, big.meas[big.lmk_idx == 2]])
    p = BAProblem(K=q.K, cam_means=q.cam_means, lmk_means=lm, meas=meas, cam_idx=cidx.astype(np.int32), lmk_idx=lidx.astype(np.int32))
    ref = BAEngine.from_problem(p)
    assert ref.info()['cam_groups'] == (1 if windows else 0) and (ref.plan_info()['max_window'] > 0) == windows, ref.plan_info()
    ref.generate_priors_var(50.0); ref.update_all_beliefs()
    ares, energies = oracle_mod.replay_ba(ref, 10, diagnostics=True)
    rce, rcl, rle, rll = ref.beliefs()
    ranks = run_world(p, 2, True, 10, oracle_mod, True)
    lo = 0
    for r in ranks:
        assert np.array_equal(r['ce'], ranks[0]['ce']) and np.array_equal(r['cl'], ranks[0]['cl'])
        assert rel_err_rows(r['ce'], rce) < 1e-6 and rel_err_rows(r['cl'], rcl) < 1e-6
        a, b = r['rng']
        assert a == lo
        lo = b
        assert rel_err_rows(r['le'], rle[a:b]) < 1e-6 and rel_err_rows(r['ll'], rll[a:b]) < 1e-6
        assert np.allclose(r['ares'], ares, rtol=1e-7)
    assert lo == p.n_lmks


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: the 1M-factor headline graph cut into 2 / 4 / 8 landmark shards.  No multi-GPU node is available to
# the build, so every rank is a thread with its own engine and stream on ONE MI355X; the sweeps run through the in-library
# loop (gbp_ba_iterate_sharded: fused sweep of the shard -> camera partial sums -> exchange -> rank-ordered finish).

FULL_SHARD_SWEEPS = 10                # the no-reset schedule bench.py times: sweep 8 relinearises every factor (gbp.py:249,72)


@pytest.fixture(scope='module')
def full_size_single():
    from gbp_amd.engine import BAEngine
    p = make_synthetic(n_cams=500, n_lmks=100_000, obs_per_lmk=10, seed=0)
    ref = BAEngine.from_problem(p)
    ref.generate_priors_var(50.0)
    ref.update_all_beliefs()
    ref.iterate(FULL_SHARD_SWEEPS)
    out = dict(p=p, bel=ref.beliefs(), are=ref.are(), energy=ref.energy(), n_relin=ref.count_relinearising(),
               relin=ref.relin_counts(FULL_SHARD_SWEEPS))
    ref.close()
    return out


def run_full_world(p, world, exchange):
    from gbp_amd.sharded import ShardedBA
    shared = LockstepWorld(world)
    out, errors = [None] * world, []

    def rank_main(r):
        try:
            import torch
            torch.cuda.set_device(0)
            g = ShardedBA(p, device=0, dist=LockstepDist(shared, r), exchange=exchange)
            assert g.library_loop and g.info()['fused']
            g.generate_priors_var(50.0)
            g.update_all_beliefs()
            g.iterate(FULL_SHARD_SWEEPS)                      # ONE C call per rank: the whole loop runs inside the library
            ce, cl = g.camera_beliefs()
            rng, le, ll = g.local_landmark_beliefs()
            out[r] = dict(ce=ce, cl=cl, le=le, ll=ll, rng=rng, are=g.are(), energy=g.energy(), F=g.F,
                          n_relin=g.count_relinearising(), relin=g.relin_counts(FULL_SHARD_SWEEPS), exchange=g.exchange)
            g.close()
        except BaseException as e:                           # noqa: BLE001 -- surface it in the main thread
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    if errors:
        raise errors[0]
    return out


@pytest.mark.parametrize('exchange', ['callback', 'peer'])
@pytest.mark.parametrize('world', [2, 4, 8])
def test_full_size_sharded_config5(full_size_single, world, exchange):
    """Camera beliefs after 10 sweeps (seven plain ones, the relinearising one, two undamped ones): bitwise identical on every rank, < 1e-6 from
    the single engine and from the REFERENCE's own beliefs (fixture G9b, sweep 10); every rank's landmark beliefs < 1e-6 from
    the single engine's; ARE / energy (normalised over all ranks) and the per-sweep relinearisation counts equal the single
    engine's.  exchange = 'callback': the all-gather is a plugged-in function (the role RCCL plays on a multi-GPU node);
    'peer': the reduce kernel stores its partial sums straight into every rank's mailbox and the finish kernel waits for the
    flags (gbp_ba_peer_connect; no collective call at all)."""
    ref = full_size_single
    p = ref['p']
    ranks = run_full_world(p, world, exchange)
    assert sum(r['F'] for r in ranks) == p.n_factors
    assert max(r['F'] for r in ranks) - min(r['F'] for r in ranks) <= 10          # balanced by factor count
    rce, rcl, rle, rll = ref['bel']
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'G9b_synthetic_full_1000000.npz')
    g9b = np.load(path) if os.path.exists(path) else None
    lo = 0
    for r in ranks:
        assert r['exchange'] == exchange
        assert np.array_equal(r['ce'], ranks[0]['ce']) and np.array_equal(r['cl'], ranks[0]['cl'])   # identical on every rank
        assert rel_err_rows(r['ce'], rce) < 1e-6 and rel_err_rows(r['cl'], rcl) < 1e-6
        a, b = r['rng']
        assert a == lo
        lo = b
        assert rel_err_rows(r['le'], rle[a:b]) < 1e-6 and rel_err_rows(r['ll'], rll[a:b]) < 1e-6
        assert r['are'] == pytest.approx(ref['are'], rel=1e-8) and r['energy'] == pytest.approx(ref['energy'], rel=1e-7)
        assert r['n_relin'] == ref['n_relin']
        assert np.array_equal(r['relin'], ref['relin'])
        if g9b is not None and f'it{FULL_SHARD_SWEEPS}_cam_eta' in g9b:
            tag = f'it{FULL_SHARD_SWEEPS}'
            assert rel_err_rows(r['ce'], g9b[tag + '_cam_eta']) < 1e-6 and rel_err_rows(r['cl'], g9b[tag + '_cam_lam']) < 1e-6
            s = g9b['lmk_sample']
            mine = (s >= a) & (s < b)
            assert rel_err_rows(r['le'][s[mine] - a], g9b[tag + '_lmk_eta'][mine]) < 1e-6
            assert rel_err_rows(r['ll'][s[mine] - a], g9b[tag + '_lmk_lam'][mine]) < 1e-6
    assert lo == p.n_lmks
    assert ref['relin'][7] > p.n_factors // 2 and not ref['relin'][:7].any()


def test_peer_exchange_times_out_instead_of_hanging(monkeypatch):
    """A rank whose peer never delivers: the finish waves give up after GBP_PEER_TIMEOUT_MS and gbp_ba_sync reports it -- the GPU is
    never left spinning."""
    from gbp_amd.engine import BAEngine
    from gbp_amd._capi import GbpError
    monkeypatch.setenv('GBP_PEER_TIMEOUT_MS', '150')
    p = make_synthetic(n_cams=20, n_lmks=400, obs_per_lmk=6, seed=1)
    a, b = BAEngine.from_problem(p), BAEngine.from_problem(p)
    handles = [a.peer_export(2, same_process=True), b.peer_export(2, same_process=True)]
    a.peer_connect(0, handles, same_process=True)
    b.peer_connect(1, handles, same_process=True)
    a.generate_priors_var(50.0)
    a.update_beliefs_sharded()                                # rank 1 never runs: its rows never arrive
    for read_back in (a.beliefs, a.are, a.means):             # nothing hands out the invalid camera beliefs meanwhile
        with pytest.raises(GbpError) as ei:
            read_back()
        assert ei.value.code == -5 and 'timed out' in str(ei.value)
    with pytest.raises(GbpError) as ei:
        a.sync()
    assert ei.value.code == -5 and 'timed out' in str(ei.value)
    a.sync()                                                  # the error is reported once
    a.close(); b.close()


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('obs', [40, 100])
def test_sharded_with_landmarks_that_span_tiles(oracle_mod, fused, obs):
    """Shards whose landmarks span tiles (dense packing at 40 factors per landmark, 100: more than a tile): the beliefs of those landmarks
    are formed by k_lmk_finish_parts on a side stream BESIDE the camera exchange, and the next sweep must wait for them.  Two ranks,
    in-library loop, against one engine on the whole graph."""
    from gbp_amd.engine import BAEngine
    p = make_synthetic(n_cams=130, n_lmks=80 if obs == 40 else 36, obs_per_lmk=obs, seed=23)
    ref = BAEngine.from_problem(p, fused=fused)
    assert ref.plan_info()['pack_mode'] == 2
    ref.generate_priors_var(50.0)
    ref.update_all_beliefs()
    ares, energies = oracle_mod.replay_ba(ref, 14, diagnostics=True)
    rce, rcl, rle, rll = ref.beliefs()
    ranks = run_world(p, 2, fused, 14, oracle_mod, True)
    lo = 0
    for r in ranks:
        assert np.array_equal(r['ce'], ranks[0]['ce']) and np.array_equal(r['cl'], ranks[0]['cl'])
        assert rel_err_rows(r['ce'], rce) < 1e-6 and rel_err_rows(r['cl'], rcl) < 1e-6
        a, b = r['rng']
        assert a == lo
        lo = b
        assert rel_err_rows(r['le'], rle[a:b]) < 1e-6 and rel_err_rows(r['ll'], rll[a:b]) < 1e-6
        assert np.allclose(r['ares'], ares, rtol=1e-7)
    assert lo == p.n_lmks


@pytest.mark.parametrize('library_loop,wave_rows', [(True, True), (False, True), (True, False)])
def test_sharded_sequence_with_camera_windows(oracle_mod, monkeypatch, library_loop, wave_rows):
    """A sequence of 1 500 cameras in two landmark shards: each rank's fused sweep runs with per-workgroup camera windows (far more cameras
    than one LDS table holds), and the merged reduce-exchange-finish launch reads a camera's rows through the same per-camera row ranges.
    Against one engine on the whole graph."""
    from gbp_amd.engine import BAEngine
    monkeypatch.delenv('GBP_WINDOWS', raising=False)
    if wave_rows:
        monkeypatch.delenv('GBP_ROWS_WAVE_MAX', raising=False)       # one wave per camera adds its few rows
    else:
        monkeypatch.setenv('GBP_ROWS_WAVE_MAX', '0')                 # the tree form, as for a camera with many rows
    p = make_synthetic(n_cams=1500, n_lmks=6000, obs_per_lmk=6, seed=31, window=14)
    ref = BAEngine.from_problem(p)
    assert ref.plan_info()['fused'] and ref.plan_info()['max_window'] > 0, ref.plan_info()
    ref.generate_priors_var(50.0)
    ref.update_all_beliefs()
    ares, energies = oracle_mod.replay_ba(ref, 14, diagnostics=True)
    rce, rcl, rle, rll = ref.beliefs()
    ranks = run_world(p, 2, None, 14, oracle_mod, library_loop)
    lo = 0
    for r in ranks:
        assert r['plan']['fused'] and r['plan']['max_window'] > 0, r['plan']
        assert np.array_equal(r['ce'], ranks[0]['ce']) and np.array_equal(r['cl'], ranks[0]['cl'])
        assert rel_err_rows(r['ce'], rce) < 1e-6 and rel_err_rows(r['cl'], rcl) < 1e-6
        a, b = r['rng']
        assert a == lo
        lo = b
        assert rel_err_rows(r['le'], rle[a:b]) < 1e-6 and rel_err_rows(r['ll'], rll[a:b]) < 1e-6
        assert np.allclose(r['ares'], ares, rtol=1e-7)
    assert lo == p.n_lmks
