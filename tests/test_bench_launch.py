"""bench.py's launch path in the build container: `python bench.py --gpus 2` WITHOUT torchrun must spawn its two ranks
itself, rendezvous on 127.0.0.1, run the sharded driver and print exactly one JSON line from rank 0.  The GPU engine cannot
run here, so the ranks drive the oracle test double over gloo through tests/tools/bench_dry_launch.py -- a wrapper around bench.main();
bench.py's own command line has no way to time anything but the HIP engine -- and the line is marked "dry_run": not a measurement."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world', [2, 4])
def test_bench_spawns_its_own_ranks(oracle_mod, world):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(REPO, 'tests', 'tools', 'bench_dry_launch.py'), '--gpus', str(world), '--steps', '3', '--warmup', '1',
           '--bal', os.path.join(REPO, 'tests', 'golden', 'data', 'fr1desk_vsmall.txt'),
           '--backend', 'gloo', '--single-batch']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == world and out['steps'] == 3 and out['dry_run'] is True
    assert out['value'] > 0 and out['scaling'] == 'strong' and out['config']['n_factors'] == 1801
    assert 0.0 <= out['roofline']['frac'] <= 1.0


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_bench_command_line_cannot_reach_the_checker():
    """nothing on bench.py's command line may put tests/ or oracle/ behind the timed region."""
    src = open(os.path.join(REPO, 'bench.py')).read()
    assert 'engine-factory' not in src and 'sys.path.insert(0, os.path.join(REPO, \'tests\'))' not in src
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--engine-factory', 'tools.shard_double:factory'],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'unrecognized arguments' in r.stderr


# ---- `--gpus N` preflight ------------------------------------------------------------------------------------

def test_preflight_logic():
    sys.path.insert(0, REPO)
    import bench
    ok = bench.preflight(4, lambda: 8, lambda i, j: True, lambda i: f'gpu{i}')
    assert ok == {"devices": 8, "ranks": 4, "shared_gpu": False, "names": ['gpu0', 'gpu1', 'gpu2', 'gpu3'], "peer_access": "all pairs"}
    with pytest.raises(bench.Preflight, match=r'--gpus 8 but only 4 device'):
        bench.preflight(8, lambda: 4, lambda i, j: True)
    with pytest.raises(bench.Preflight, match=r'devices 1 and 3 cannot access each other'):
        bench.preflight(4, lambda: 4, lambda i, j: {i, j} != {1, 3})
    shared = bench.preflight(8, lambda: 1, lambda i, j: False, lambda i: 'one', share_gpu=True)      # one-GPU test boxes: nothing between devices to check
    assert shared['shared_gpu'] is True and shared['devices'] == 1 and shared['ranks'] == 8
    with pytest.raises(bench.Preflight):
        bench.preflight(2, lambda: 0, lambda i, j: True, share_gpu=True)


@pytest.mark.parametrize('under_torchrun', [False, True], ids=['launcher', 'rank0-of-torchrun'])
def test_a_node_that_cannot_run_the_job_gets_an_error_line_not_a_traceback(under_torchrun):
    """No GPU here: `bench.py --gpus 2` must say so in ONE JSON line with "error" and exit with the preflight code -- from the launcher
    before anything is spawned, and from rank 0 when torchrun did the spawning -- instead of the traceback of whichever rank died first."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    if under_torchrun:
        env.update(RANK='0', WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['value'] is None and out['n_gpus'] == 2 and 'no GPU visible' in out['error'] and 'Traceback' not in r.stderr
