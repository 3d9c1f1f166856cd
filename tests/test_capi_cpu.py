"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, exports every symbol
include/gbp_ba.h declares, and refuses to run without a GPU (no CPU fallback in the product path)."""
import ctypes as ct
import os
import re

import numpy as np
import pytest

from conftest import DATA, REPO


@pytest.fixture(scope='module')
def capi():
    from gbp_amd import build, _capi
    build.build()
    return _capi


def declared_symbols():
    import glob
    names = set()
    for h in glob.glob(os.path.join(REPO, 'include', '*.h')):
        text = re.sub(r'/\*.*?\*/', '', open(h).read(), flags=re.S)
        names |= set(re.findall(r'\b(gbp_[a-z_0-9]+)\s*\(', text))
    return sorted(names)


def test_header_symbols_all_exported_and_bound(capi):
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported by libgbp_hip.so"
    assert set(names) == set(capi.SIGNATURES), set(names) ^ set(capi.SIGNATURES)
    assert lib.gbp_abi_version() == capi.ABI_VERSION == 3


def test_desc_struct_matches_header_layout(capi):
    # 4 int32, 4 doubles, 5 pointers, 1 double, 4 int32, 3 doubles on LP64
    assert ct.sizeof(capi.Desc) == 16 + 32 + 40 + 8 + 16 + 24


def test_product_path_has_no_cpu_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from gbp_amd.engine import BAEngine
    with pytest.raises(capi.GbpError) as ei:
        BAEngine([500., 500., 320., 240.], np.zeros((1, 6)), np.zeros((1, 3)), np.zeros((1, 2)), [0], [0])
    assert ei.value.code == -4 and 'no CPU path' in str(ei.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, 'gbp_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.cpp', '.h')):
                src = open(os.path.join(root, f)).read()
                assert 'oracle' not in src.replace('gbp_oracle.c header', ''), f"{f} mentions the oracle"


def test_bad_arguments_are_reported_not_crashed(capi):
    lib = capi.load()
    h = ct.c_void_p()
    assert lib.gbp_ba_create(ct.byref(h), None) == -1
    d = capi.Desc()
    d.n_factors = -1
    assert lib.gbp_ba_create(ct.byref(h), ct.byref(d)) == -1
    assert b'negative' in lib.gbp_last_error()
    assert lib.gbp_ba_iterate(None, 1, 1, 1) == -1


def test_native_bal_reader_equals_python_reader(tmp_path):
    """gbp_bal_header / gbp_bal_read (host-only C++) against the Python restatement of utils/read_balfile.py:4-37:
    identical arrays on every data file of the reference and on a generated file; malformed files are refused."""
    import glob
    from gbp_amd import _capi
    from gbp_amd.balio import read_bal, read_bal_native
    from gbp_amd.synthetic import make_synthetic, write_bal
    _capi.load()
    files = sorted(glob.glob(os.path.join(DATA, '*.txt')))
    assert files
    syn = os.path.join(tmp_path, 'syn.txt')
    write_bal(make_synthetic(n_cams=12, n_lmks=300, obs_per_lmk=5, seed=3), syn)
    for f in files + [syn]:
        a, b = read_bal(f, native=False), read_bal_native(f)
        for k in ('K', 'cam_means', 'lmk_means', 'meas', 'cam_idx', 'lmk_idx'):
            x, y = getattr(a, k), getattr(b, k)
            assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y), (f, k)
    bad = os.path.join(tmp_path, 'bad.txt')
    with open(syn) as fh:
        lines = fh.read().split('\n')
    with open(bad, 'w') as fh:
        fh.write('\n'.join(lines[:40]))                       # truncated
    with pytest.raises(_capi.GbpError):
        read_bal_native(bad)
    with pytest.raises(_capi.GbpError):
        read_bal_native(os.path.join(tmp_path, 'missing.txt'))
