"""World-size-2 run of the landmark-sharded sweep's HOST logic (gbp_amd/sharded.py) over gloo on CPU.

The HIP engine cannot run here, so each rank drives oracle.OracleShard (a test double with the same five calls the
driver makes on BAEngine).  Checks: partition covers every factor exactly once, the rank-ordered camera exchange
reproduces the single-process result, diagnostics are globally normalised, both ranks hold identical camera beliefs.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import DATA, REPO, rel_err_rows


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


from tools.shard_double import OracleAdapter as _OracleAdapter   # noqa: E402  (tests/ is on sys.path via conftest)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gbp_amd.balio import read_bal
    from gbp_amd.sharded import ShardedBA
    from oracle import oracle
    p = read_bal(os.path.join(DATA, 'fr1desk_small.txt'))
    g = ShardedBA(p, engine_factory=_OracleAdapter)
    g.generate_priors_var(50.0)
    g.update_all_beliefs()
    ares, _ = oracle.replay_ba(g, 12, diagnostics=True)
    ce, cl = g.camera_beliefs()
    (lo, hi), le, ll = g.local_landmark_beliefs()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), ce=ce, cl=cl, le=le, ll=ll, lo=lo, hi=hi, ares=ares,
             F=g.F, energy=g.energy())
    dist.barrier()
    dist.destroy_process_group()


def _worker_local(rank, world, port, out_dir):
    """the same job, every rank handing over ITS shard only (ShardedBA(local_shard=True))"""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gbp_amd.balio import read_bal
    from gbp_amd.sharded import ShardedBA, local_problem, partition_landmarks
    from oracle import oracle
    p = read_bal(os.path.join(DATA, 'fr1desk_small.txt'))
    b = partition_landmarks(p.lmk_idx, p.n_lmks, world)
    mine = local_problem(p, int(b[rank]), int(b[rank + 1]))
    del p
    g = ShardedBA(mine, engine_factory=_OracleAdapter, local_shard=True)
    g.generate_priors_var(50.0)
    g.update_all_beliefs()
    ares, _ = oracle.replay_ba(g, 12, diagnostics=True)
    ce, cl = g.camera_beliefs()
    (lo, hi), le, ll = g.local_landmark_beliefs()
    np.savez(os.path.join(out_dir, f'local{rank}.npz'), ce=ce, cl=cl, le=le, ll=ll, lo=lo, hi=hi, ares=ares, F=g.F, F_total=g.F_total, L_total=g.L_total)
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    from gbp_amd.balio import read_bal
    from gbp_amd.sharded import partition_landmarks, local_problem
    p = read_bal(os.path.join(DATA, 'fr1desk.txt'))
    for world in (1, 2, 3, 8):
        b = partition_landmarks(p.lmk_idx, p.n_lmks, world)
        assert b[0] == 0 and b[-1] == p.n_lmks and np.all(np.diff(b) >= 0)
        sizes = [local_problem(p, int(b[r]), int(b[r + 1])).n_factors for r in range(world)]
        assert sum(sizes) == p.n_factors
        assert max(sizes) - min(sizes) <= 2 * np.bincount(p.lmk_idx).max()
    # degenerate: more ranks than landmarks with factors
    b = partition_landmarks(np.array([0, 0, 1]), 2, 4)
    assert b[0] == 0 and b[-1] == 2 and len(b) == 5


@pytest.mark.timeout(300)
def test_two_rank_sweep_matches_single_process(tmp_path, oracle_mod):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(os.path.join(tmp_path, f'rank{r}.npz')) for r in range(world))
    # every rank ends with bitwise identical camera beliefs (rank-ordered sum on every rank)
    assert np.array_equal(r0['ce'], r1['ce']) and np.array_equal(r0['cl'], r1['cl'])
    assert np.array_equal(r0['ares'], r1['ares']) and r0['energy'] == r1['energy']
    from gbp_amd.balio import read_bal
    p = read_bal(os.path.join(DATA, 'fr1desk_small.txt'))
    assert int(r0['F']) + int(r1['F']) == p.n_factors
    o = oracle_mod.OracleBA.from_problem(p)
    o.generate_priors_var(50.0)
    o.update_all_beliefs()
    ares, _ = oracle_mod.replay_ba(o, 12, diagnostics=True)
    ce, cl, le, ll = o.beliefs()
    assert rel_err_rows(r0['ce'], ce) < 1e-7 and rel_err_rows(r0['cl'], cl) < 1e-7
    le2 = np.concatenate([r0['le'], r1['le']])
    ll2 = np.concatenate([r0['ll'], r1['ll']])
    assert int(r0['lo']) == 0 and int(r0['hi']) == int(r1['lo']) and int(r1['hi']) == p.n_lmks
    assert rel_err_rows(le2, le) < 1e-7 and rel_err_rows(ll2, ll) < 1e-7
    assert np.allclose(r0['ares'], ares, rtol=1e-8)
    assert float(r0['energy']) == pytest.approx(o.energy(), rel=1e-7)


@pytest.mark.timeout(300)
def test_ranks_that_bring_their_own_shard(tmp_path, oracle_mod):
    """ShardedBA(local_shard=True): every rank hands over its own landmarks + the shared cameras (bench.py's secondary workload makes
    them on the rank).  Cut from the same file by the same partition, the job must be the one the whole-problem constructor runs: bitwise."""
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    mp.spawn(_worker_local, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        a, b = np.load(os.path.join(tmp_path, f'rank{r}.npz')), np.load(os.path.join(tmp_path, f'local{r}.npz'))
        for k in ('ce', 'cl', 'le', 'll', 'ares'):
            assert np.array_equal(a[k], b[k]), (r, k)
        assert (int(a['lo']), int(a['hi']), int(a['F'])) == (int(b['lo']), int(b['hi']), int(b['F']))
        assert int(b['F_total']) == 3917 and int(b['L_total']) > 0
