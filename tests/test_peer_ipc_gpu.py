"""The peer-store camera exchange between real PROCESSES (gbp_ba_peer_export / gbp_ba_peer_connect with hipIpc handles): two
and three ranks, each its own process, all on the one GPU this box has.  What the thread-rank tests cannot show: mailboxes mapped
across address spaces, kernels of different processes that genuinely wait for each other's stores (no rendezvous on the host:
the merged reduce-exchange-finish launch polls the row tags), the time-out instead of a hang when a rank never arrives."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, rel_err_rows
from gbp_amd.synthetic import make_synthetic

pytestmark = pytest.mark.gpu
WORKER = os.path.join(REPO, 'tests', 'peer_ipc_worker.py')


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(tmp_path, world, n_sweeps, n_cams, n_lmks, extra_env=None, expect_fail=False):
    port = free_port()
    # The ranks SHARE one GPU here: a rank that waits inside its exchange launch holds LDS on the CUs it sits on, and another rank's
    # fused sweep needs a whole CU's LDS per workgroup -- with the production grid (128 workgroups) two waiting ranks can leave the
    # third no CU at all until the driver time-slices the processes.  16 waiting workgroups per rank leave room.  (One rank per GPU,
    # the real deployment, has no such coupling.)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GBP_XCHG_BLOCKS='16', **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), str(tmp_path), str(n_sweeps), str(n_cams), str(n_lmks)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors='replace'))
    if expect_fail:
        return outs
    for pr, out in zip(procs, outs):
        assert pr.returncode == 0, out[-3000:]
    return [np.load(os.path.join(tmp_path, f'rank{r}.npz')) for r in range(world)]


@pytest.mark.parametrize('world,n_cams,n_lmks', [(2, 60, 4000), (3, 500, 20000), (4, 500, 100000)])
def test_peer_exchange_between_processes(tmp_path, world, n_cams, n_lmks):
    """(4, 500, 100000) is BASELINE configs[4] at full size -- the 1M-factor graph over four ranks -- between real processes: 12 sweeps
    through the one that relinearises every factor."""
    from gbp_amd.engine import BAEngine
    n_sweeps = 12
    p = make_synthetic(n_cams=n_cams, n_lmks=n_lmks, obs_per_lmk=10, seed=2)
    ref = BAEngine.from_problem(p)
    ref.generate_priors_var(50.0)
    ref.update_all_beliefs()
    ref.iterate(n_sweeps)
    rce, rcl, rle, rll = ref.beliefs()
    are = ref.are()
    ref.close()
    ranks = run_ranks(tmp_path, world, n_sweeps, n_cams, n_lmks)
    assert sum(int(r['F']) for r in ranks) == p.n_factors
    lo = 0
    for r in ranks:
        assert str(r['kind']) == 'peer' and int(r['n_ranks']) == world
        assert np.array_equal(r['ce'], ranks[0]['ce']) and np.array_equal(r['cl'], ranks[0]['cl'])   # identical on every rank
        assert rel_err_rows(r['ce'], rce) < 1e-6 and rel_err_rows(r['cl'], rcl) < 1e-6
        a, b = int(r['lo']), int(r['hi'])
        assert a == lo
        lo = b
        assert rel_err_rows(r['le'], rle[a:b]) < 1e-6 and rel_err_rows(r['ll'], rll[a:b]) < 1e-6
        assert float(r['are']) == pytest.approx(are, rel=1e-8)
    assert lo == p.n_lmks


@pytest.mark.parametrize('world,n_cams,n_lmks', [(2, 60, 4000), (3, 700, 9000)])
def test_peer_exchange_general_sweep_between_processes(tmp_path, world, n_cams, n_lmks):
    """The general sweep under the peer-store exchange: everything behind the factor kernel is ONE launch (k_cam_staged_xchg: staged rows ->
    camera sums -> mailboxes of all ranks -> wait -> rank-ordered finish), between real processes; 700 cameras do not fit the fused
    sweep's table at all.  Against one engine on the whole graph."""
    from gbp_amd.engine import BAEngine
    n_sweeps = 12
    p = make_synthetic(n_cams=n_cams, n_lmks=n_lmks, obs_per_lmk=10, seed=2)
    ref = BAEngine.from_problem(p, fused=False)
    ref.generate_priors_var(50.0)
    ref.update_all_beliefs()
    ref.iterate(n_sweeps)
    rce, rcl, rle, rll = ref.beliefs()
    are = ref.are()
    ref.close()
    ranks = run_ranks(tmp_path, world, n_sweeps, n_cams, n_lmks, extra_env={'GBP_TEST_FUSED': '0'})
    for r in ranks:
        assert str(r['kind']) == 'peer' and int(r['n_ranks']) == world
        assert np.array_equal(r['ce'], ranks[0]['ce']) and np.array_equal(r['cl'], ranks[0]['cl'])
        assert rel_err_rows(r['ce'], rce) < 1e-6 and rel_err_rows(r['cl'], rcl) < 1e-6
        a, b = int(r['lo']), int(r['hi'])
        assert rel_err_rows(r['le'], rle[a:b]) < 1e-6 and rel_err_rows(r['ll'], rll[a:b]) < 1e-6
        assert float(r['are']) == pytest.approx(are, rel=1e-8)


def test_peer_exchange_split_launches_between_processes(tmp_path):
    """GBP_PEER_SPLIT=1: reduce / push and the waiting finish as two launches (what the general sweep and update_all_beliefs always
    use), also across processes."""
    ranks = run_ranks(tmp_path, 2, 6, 40, 2000, extra_env={'GBP_PEER_SPLIT': '1'})
    assert np.array_equal(ranks[0]['ce'], ranks[1]['ce'])


@pytest.mark.parametrize('world', [2, 4, 8])
def test_bench_n_ranks_on_one_gpu(tmp_path, world):
    """bench.py's N > 1 path on hardware, as far as a one-GPU box allows: `--gpus N` spawns its N ranks (torchrun on 127.0.0.1), all
    on device 0 (GBP_BENCH_SHARE_GPU), side channel gloo, camera exchange = peer stores between the processes (after the self-test of
    gbp_ba_peer_selftest).  The line must carry every field a SCALE line is judged by: every rank's device times and its own roofline,
    the rank count the exchange reports, the parity check against the reference's own run of this graph.  (Its `value` is N ranks
    time-slicing one GPU: not a measurement of anything -- `one_rank_per_device` is false and says so.)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(GBP_BENCH_SHARE_GPU='1', GBP_XCHG_BLOCKS='16', HSA_ENABLE_IPC_MODE_LEGACY='0', GBP_PEER_TIMEOUT_MS='20000')
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(world), '--steps', '20', '--warmup', '5',
           '--backend', 'gloo', '--exchange', 'peer', '--single-batch', '--secondary-lmks', str(8000 * world)]      # the headline graph: parity_check has its fixture
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    if os.environ.get('GBP_KEEP_BENCH_LINES'):               # (tools/profile_round.sh keeps the lines as profiles/r05_bench_gpusN_shared_gpu_sample.json)
        with open(os.path.join(os.environ['GBP_KEEP_BENCH_LINES'], f'bench_gpus{world}_shared_gpu.json'), 'w') as f:
            json.dump(out, f, indent=1)
    cfg = out['config']
    assert out['n_gpus'] == world and out['value'] > 0 and out['scaling'] == 'strong'
    assert cfg['exchange'] == 'peer' and cfg['exchange_requested'] == 'peer' and cfg['exchange_fallback'] is None and cfg['loop'] == 'in-library'
    assert len(out['per_rank']) == world and all(pr['ranks_reported_by_exchange'] == world for pr in out['per_rank'])
    assert sum(pr['n_factors'] for pr in out['per_rank']) == 1_000_000 and sum(pr['n_lmks'] for pr in out['per_rank']) == 100_000
    for pr in out['per_rank']:                                # every rank's own roofline: its shard's layout bytes / its sweep kernel's time
        rf = pr['roofline']
        assert pr['sweep'] in ('fused', 'general') and rf['bound'] == 'hbm' and rf['bytes_per_launch'] > 0.9 * pr['n_factors'] * 256
        assert 0.0 < rf['frac'] <= 1.0 and rf['kernel_avg_ms'] > 0 and abs(rf['achieved'] - rf['bytes_per_launch'] / rf['kernel_avg_ms'] / 1e6) < 1e-6 * rf['achieved']
        assert pr['sweep_ms'] > 0 and pr['reduce_and_exchange_ms'] > 0 and pr['step_ms_device'] > 0
    # the line proves itself: ten sweeps outside the timed region against the REFERENCE's own run of this graph (fixture G9b)
    pc = out['parity_check']
    assert pc['ok'] is True and pc['camera_beliefs_bitwise_equal_across_ranks'] is True and pc['ranks_reported_by_exchange'] == world
    assert pc['camera_belief_gap_vs_reference'] < 1e-6 and pc['are_trace_max_rel_err'] < 1e-6 and len(pc['are_trace']) == 11
    assert pc['distinct_devices'] == 1 and pc['one_rank_per_device'] is False      # (the ranks share this box's one GPU, and the line says so)
    # the node's topology, the stated model of the step, and the bandwidth-bound companion travel in the line
    topo = cfg['topology']
    assert topo['shared_gpu'] is True and topo['ranks'] == world and topo['devices'] == 1
    pred = out['prediction']
    assert out['predicted_step_ms'] == pred['predicted_step_ms'] and 0.005 < pred['predicted_step_ms'] < 1.0
    assert len(pred['solo_step_ms_per_rank']) == world and max(pred['solo_step_ms_per_rank']) == pred['predicted_step_ms']
    assert pred['measured_step_ms'] == pytest.approx(out['ms_per_step'], rel=1e-9)
    sec = out['secondary']
    assert sec['label'].startswith('SECONDARY') and sec['n_factors'] == 80_000 * world and sec['value'] > 0 and sec['exchange'] == 'peer'
    assert len(sec['per_rank']) == world and all(pr['n_factors'] == 80_000 and pr['ranks_reported_by_exchange'] == world for pr in sec['per_rank'])


def test_bench_line_survives_an_rccl_that_does_not_come_up(tmp_path):
    """`--exchange rccl` on a node whose RCCL fails at init (GBP_RCCL_FAIL: gbp_ba_comm_unique_id refuses on rank 0, the
    refusal travels to every rank as data).  The job must not die: ShardedBA falls back -- here, with a gloo side channel, to the
    Python-driven loop -- and the ONE line says which exchange ran and why the requested one did not."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(GBP_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0', GBP_RCCL_FAIL='1')
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '10', '--warmup', '2', '--backend', 'gloo', '--exchange', 'rccl',
           '--single-batch', '--secondary-lmks', '0']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    cfg = out['config']
    assert cfg['exchange_requested'] == 'rccl' and cfg['exchange'] == 'python' and cfg['loop'] == 'python'
    assert 'GBP_RCCL_FAIL' in cfg['exchange_fallback'], cfg
    assert out['value'] > 0 and out['parity_check']['ok'] is True and out['parity_check']['camera_beliefs_bitwise_equal_across_ranks'] is True


def test_peer_selftest_names_the_pair_that_failed(tmp_path):
    """gbp_ba_peer_selftest: a rank whose peer never sends its probe row (here: rank 1 connects but skips its self-test) gets GBP_ESTATE
    naming the rank that did not arrive, within the given time -- not a hang, and not a wrong belief later."""
    outs = run_ranks(tmp_path, 2, 2, 40, 2000, extra_env={'GBP_TEST_SELFTEST_ONLY': '1'}, expect_fail=True)
    assert 'SELFTEST-FAILED' in outs[0] and 'probe row of rank 1 did not reach rank 0 within 300 ms' in outs[0], outs[0][-1500:]
    assert 'SELFTEST-SKIPPED' in outs[1]
