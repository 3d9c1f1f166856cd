"""The history holds sources, fixtures and measured profiles -- never compiler output (code objects of an
earlier library had been committed beside the product and shipped to the GPU box looking like product binaries)."""
import os
import subprocess

import pytest

from conftest import REPO

BINARY_OK = ('tests/golden/',)                 # .npz fixtures: arrays generated from the reference (tests/golden/make_*.py)
BINARY_MAGIC = (b'\x7fELF', b'__CLANG_OFFLOAD_BUNDLE__', b'BC\xc0\xde', b'!<arch>\n')


def tracked_files():
    if not os.path.isdir(os.path.join(REPO, '.git')):
        pytest.skip("not a git checkout (the GPU box gets a snapshot)")
    try:
        out = subprocess.run(['git', 'ls-files', '-z'], cwd=REPO, capture_output=True, check=True).stdout
    except (OSError, subprocess.CalledProcessError) as e:
        pytest.skip(f"git ls-files failed: {e}")
    return [f for f in out.decode().split('\0') if f]


def test_no_build_products_are_tracked():
    bad = []
    for f in tracked_files():
        path = os.path.join(REPO, f)
        if f.startswith(BINARY_OK) or not os.path.isfile(path):
            continue
        with open(path, 'rb') as fh:
            head = fh.read(4096)
        if head.startswith(BINARY_MAGIC) or b'\0' in head:
            bad.append(f)
        if os.path.getsize(path) == 0 and not f.endswith('__init__.py'):
            bad.append(f + ' (empty)')
    assert not bad, f"compiler output / binary files are tracked: {bad}"


def test_ignore_files_lock_compiler_byproducts_out():
    for name in ('.gitignore', '.gpurunignore'):
        pats = open(os.path.join(REPO, name)).read().split()
        for p in ('*.hipfb', '*.hipv4-*', '*.host-x86_64-*', 'gbp_amd/libgbp_hip.so.*'):
            assert p in pats, f"{name} lacks {p}"
    # ... while the product library itself must keep travelling to the GPU box
    assert '*.so' not in open(os.path.join(REPO, '.gpurunignore')).read().split()
