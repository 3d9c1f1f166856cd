"""Build libgbp_hip.so (the HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so ships
to the GPU box with the source snapshot (it is git-ignored, not gpurun-ignored).

    python -m gbp_amd.build [--out PATH] [-DNAME[=VALUE] ...] [--flag=-Rpass-analysis=...]

`--out` / `-D` build a scratch copy of the library (tools/profile_round.sh: -DGBP_FUSED_DBG_SWITCHES, tools/phase_profile.py:
-DGBP_PHASE_TIMING) and leave the product library alone.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libgbp_hip.so')
# one translation unit per part of the C ABI (gbp_handle.hpp says which), compiled side by side and linked into ONE library
SOURCES = ['gbp_capi.hip', 'gbp_capi_sweep.hip', 'gbp_capi_shard.hip', 'gbp_capi_views.hip', 'gbp_capi_state.hip', 'gbp_lin_capi.hip',
           'gbp_sort.hip']
HEADERS = ['gbp_handle.hpp', 'gbp_build.hpp', 'gbp_kernels.hpp', 'gbp_sweep_kernels.hpp', 'gbp_view_kernels.hpp', 'gbp_fused.hpp',
           'gbp_fused_plan.hpp', 'gbp_math.hpp', 'gbp_balio.hpp', os.path.join('experimental', 'gbp_instrument.hpp'),
           os.path.join('..', '..', 'include', 'gbp_ba.h'), os.path.join('..', '..', 'include', 'gbp_lin.h')]
DEPS = SOURCES + HEADERS
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast']


def hipcc_path():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libgbp_hip.so cannot be built (there is no CPU fallback)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, out=None, defines=(), extra_flags=(), capture=False):
    """The same command lines every time: the library's sha256 ties committed counter passes to the binary they measured (bench.py),
    and even a remarks flag changes it (tools/resource_usage.py compiles a scratch copy for that: out=..., extra_flags=...).
    Returns the library's path (capture=True: (path, the compiler's stderr))."""
    lib = out or LIB
    if not force and out is None and not needs_build():
        return (lib, '') if capture else lib
    hipcc = hipcc_path()
    flags = FLAGS + [f'-D{d}' for d in defines] + list(extra_flags)
    log = []
    # a FIXED object directory (git-ignored `build/`): object paths end up inside the library, and the library's sha256 must be the same
    # wherever and whenever these sources are built (bench.py ties committed counter passes to it)
    tmp = os.path.join(HERE, 'build', 'obj_' + hashlib.sha256(' '.join(flags).encode()).hexdigest()[:12])
    os.makedirs(tmp, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(tmp, os.path.splitext(src)[0] + '.o')
        r = subprocess.run([hipcc] + flags + ['-c', src, '-o', os.path.relpath(obj, CSRC)], cwd=CSRC, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-6000:]}")
        log.append(r.stderr)
        return obj

    # Two builds with the same flags share the object directory (xdist workers that both see needs_build(), a tool's scratch build beside
    # the product's): one at a time, under a lock on the directory; and the library appears under its name only when it is complete
    # (linked beside it, then os.replace): nobody can dlopen a half-written libgbp_hip.so.
    with open(os.path.join(tmp, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and out is None and not needs_build():      # somebody else built it while this process waited
            return (lib, '') if capture else lib
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        partial = f'{lib}.link{os.getpid()}.tmp'
        try:
            # (the output NAME is not recorded inside the library: the sha256 is that of a link straight to `lib`)
            r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', partial] + [os.path.relpath(o, CSRC) for o in objs], cwd=CSRC,
                               capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"linking {lib} failed:\n{r.stderr[-6000:]}")
            os.replace(partial, lib)
        finally:
            if os.path.exists(partial):
                os.remove(partial)
    return (lib, ''.join(log)) if capture else lib


if __name__ == '__main__':
    out, defines, extra = None, [], []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == '--out':
            out = os.path.abspath(args.pop(0))
        elif a.startswith('-D'):
            defines.append(a[2:])
        elif a.startswith('--flag='):
            extra.append(a[len('--flag='):])
        else:
            raise SystemExit(f"unknown argument {a!r}")
    print(build(force=True, out=out, defines=defines, extra_flags=extra))
