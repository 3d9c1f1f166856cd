"""AddressSanitizer + UndefinedBehaviorSanitizer over what CAN be sanitised on a CPU:
the C oracle (oracle/gbp_oracle.c) and the sweep kernels' own per-factor C++ compiled for the host (tests/hostmath/host_math.hip =
gbp_math.hpp + factor_core of gbp_kernels.hpp), both rebuilt with -fsanitize=address,undefined (-fno-sanitize-recover: any finding ends
the process) into a scratch directory and driven through whole schedules by the ordinary tests, in a child interpreter that has the
sanitizer runtime preloaded.  The schedules cover ba.py's 30-sweep replay on a data file (two relinearisation waves), a robust loss,
prior weakening, the dense remainder and the relinearisation clock's wrap-around -- every branch of factor_core.

(The device side cannot be sanitised here: there is no GPU in the build container and the ASan device runtime needs xnack.  Its
lock-free protocols are pinned on the GPU box instead: the same-address ds_add_f64 lane order by k_single_probe -- at create, in
test_edge_shapes_gpu.py and in __graft_entry__.smoke() -- and the mailbox hand-off by gbp_ba_peer_selftest.)"""
import glob
import os
import shutil
import subprocess
import sys

import pytest

from conftest import REPO

CLANG = '/opt/rocm/lib/llvm/bin/clang'
SAN = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-shared-libsan', '-g', '-O1', '-fPIC', '-shared']


def asan_runtime():
    hits = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    return hits[0] if hits else None


@pytest.fixture(scope='module')
def sanitized(tmp_path_factory):
    rt = asan_runtime()
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if rt is None or not os.path.exists(CLANG) or not os.path.exists(hipcc):
        pytest.skip("ROCm clang / its ASan runtime not found")
    d = tmp_path_factory.mktemp('san')
    ora, hm = str(d / 'libgbp_oracle_san.so'), str(d / 'libhostmath_san.so')
    # the oracle: same flags as oracle/Makefile minus OpenMP (one thread: the sanitizers see every access in program order)
    subprocess.check_call([CLANG, '-std=c11', '-ffp-contract=off', '-Wall', '-Wextra'] + SAN + ['-o', ora, os.path.join(REPO, 'oracle', 'gbp_oracle.c'), '-lm'])
    # the kernels' per-factor C++: host pass only (the device pass cannot be sanitised without a GPU)
    subprocess.check_call([hipcc, '--offload-host-only', '-std=c++17', '-ffp-contract=fast'] + SAN + ['-o', hm, os.path.join(REPO, 'tests', 'hostmath', 'host_math.hip')])
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1', GBP_ORACLE_SANITIZED_LIB=ora, GBP_HOSTMATH_SANITIZED_LIB=hm, OMP_NUM_THREADS='1')
    return env


def run_child(env, args):
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider'] + args, cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert 'ERROR: AddressSanitizer' not in tail and 'runtime error:' not in tail, tail
    assert r.returncode == 0, tail
    return r.stdout


def test_sanitizer_runtime_is_really_in_the_child(sanitized):
    """the child must run the SANITIZED libraries: a test that quietly loaded the plain ones would prove nothing"""
    code = ("import ctypes, os; from oracle import oracle; L = oracle.lib(); "
            "print('MAPS', sum(('libgbp_oracle_san' in l) or ('libclang_rt.asan' in l) for l in open('/proc/self/maps')))")
    r = subprocess.run([sys.executable, '-c', code], cwd=REPO, env=sanitized, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert int(r.stdout.split('MAPS')[1]) >= 2, r.stdout


def test_host_compiled_factor_core_and_oracle_under_asan_ubsan(sanitized):
    out = run_child(sanitized, ['tests/test_factor_math_host.py', '-k',
                                'fr2robot2 or huber or prior_weakening or damped_in_the_relinearising or clock_wraps'])
    assert ' passed' in out and 'failed' not in out, out


def test_oracle_golden_vectors_under_asan_ubsan(sanitized):
    # the oracle alone through the reference's own fixtures (the -k picks 18 of the 23 fixture tests: G1, G1b, G3, G4, G7, G10-G16)
    out = run_child(sanitized, ['tests/test_oracle_golden.py', '-k', 'G1 or G3 or G4 or G7 or g1 or g3 or g4 or g7'])
    assert ' passed' in out and 'failed' not in out, out
