#!/usr/bin/env python3
"""Where the waves of the fused sweep spend their time: builds libgbp_hip.so with -DGBP_PHASE_TIMING into tools/ (the product
build has none of this code), runs the 1M-factor graph and prints, per phase of the persistent loop, the share of the
wave-time (mean over the 2048 waves of the last sweep).  Run on the GPU box.

    python tools/phase_profile.py [--build-only]
"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(HERE)
LIB = os.environ.get('PHASE_LIB', os.path.join(HERE, 'libgbp_phase.so'))
EXTRA = os.environ.get('PHASE_DEFS', '').split()
NAMES = ['ticket+descriptor', 'issue stream loads', 'lmk beliefs of prev tile (LDS)', 'wait streams', 'camera gather', 'lmk records via LDS',
         'maths', 'stores issued', 'wait accumulation turn', 'accumulate + loop', 'wait for the other waves at the end', 'table write-out']
if not os.path.exists(LIB) or '--build-only' in sys.argv:
    sys.path.insert(0, REPO)
    from gbp_amd import build
    build.build(force=True, out=LIB, defines=['GBP_PHASE_TIMING'] + [d[2:] for d in EXTRA if d.startswith('-D')],
                extra_flags=[d for d in EXTRA if not d.startswith('-D')])
    if '--build-only' in sys.argv:
        sys.exit(0)
os.environ['GBP_HIP_LIB'] = LIB
sys.path.insert(0, REPO)
import ctypes as ct
import numpy as np
from gbp_amd import _capi
from gbp_amd.engine import BAEngine
from gbp_amd.synthetic import make_synthetic
if os.environ.get('BAL'):                      # a data file instead (BAL=tests/golden/data/fr1desk.txt)
    from gbp_amd.balio import read_bal
    p = read_bal(os.environ['BAL'])
else:
    p = make_synthetic(n_cams=int(os.environ.get('CAMS', 500)), n_lmks=int(os.environ.get('LMKS', 100_000)), obs_per_lmk=int(os.environ.get('OBS', 10)), seed=0)
e = BAEngine.from_problem(p)
e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(60); e.sync()
nr, nc = ct.c_int32(), ct.c_int32()
_capi.check(e._lib.gbp_ba_phase_profile(e._h, None, 0, ct.byref(nr), ct.byref(nc)))
out = np.zeros((nr.value, nc.value), dtype=np.uint64)
_capi.check(e._lib.gbp_ba_phase_profile(e._h, out.ctypes.data_as(ct.c_void_p), nr.value, ct.byref(nr), ct.byref(nc)))
tot = out.sum(axis=1).astype(np.float64)
busy = out[:, 6] > 0                            # waves that had a tile (small graphs: one wave per workgroup)
print(f"waves with a tile: {int(busy.sum())} of {nr.value}; their ticks per phase (mean):", [int(x) for x in out[busy].astype(np.float64).mean(axis=0)])
print(f"waves {nr.value}; ticks per wave: mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f} (s_memtime ticks)")
share = out.astype(np.float64).sum(axis=0) / tot.sum()
for n, s, m in zip(NAMES, share, out.astype(np.float64).mean(axis=0)):
    print(f"  {n:34s} {100 * s:5.1f} %   {m:9.0f} ticks/wave")
# per-workgroup finishing time (its slowest wave): the kernel lasts as long as the slowest workgroup
nb = ct.c_int32(); _capi.check(e._lib.gbp_ba_info(e._h, None, None, ct.byref(nb)))
wg = tot.reshape(nb.value, -1).max(axis=1)
print(f"workgroups {wg.size}: finishing ticks mean {wg.mean():.0f} min {wg.min():.0f} max {wg.max():.0f}  (max / mean = {wg.max() / wg.mean():.3f})")
for x in range(8):
    s = wg[x::8]
    print(f"  workgroups = {x} mod 8 (one XCD): mean {s.mean():.0f} min {s.min():.0f} max {s.max():.0f}")
order = np.argsort(wg)
print("  slowest workgroups:", [(int(b), int(wg[b])) for b in order[-8:]], " fastest:", [(int(b), int(wg[b])) for b in order[:4]])
