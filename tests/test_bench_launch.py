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
