"""Reader for the reference's BAL-style text files (layout: data/README.md:5-14).

Accepts what utils/read_balfile.py:4-37 accepts: leading blank lines and lines whose first token
is '#' are skipped; then `C L F`, `fx fy cx cy`, F observation rows `cam lmk u v`, then 6*C and
3*L scalars, one per line (only the first token of those lines is read).
"""
from __future__ import annotations

import numpy as np

from .synthetic import BAProblem


def read_bal_native(path) -> BAProblem:
    """The same file through the library's C++ reader (include/gbp_ba.h: gbp_bal_header / gbp_bal_read), ~30x faster."""
    import ctypes as ct
    from . import _capi
    lib = _capi.load()
    c, l, f = ct.c_int32(), ct.c_int32(), ct.c_int32()
    bpath = str(path).encode()
    _capi.check(lib.gbp_bal_header(bpath, ct.byref(c), ct.byref(l), ct.byref(f)))
    C, L, F = c.value, l.value, f.value
    K, cam, lmk, meas = np.empty(4), np.empty((C, 6)), np.empty((L, 3)), np.empty((F, 2))
    ci, li = np.empty(F, np.int32), np.empty(F, np.int32)
    _capi.check(lib.gbp_bal_read(bpath, C, L, F, _capi.dptr(K), _capi.dptr(cam), _capi.dptr(lmk), _capi.dptr(meas),
                                 _capi.iptr(ci), _capi.iptr(li)))
    return BAProblem(K=K, cam_means=cam, lmk_means=lmk, meas=meas, cam_idx=ci, lmk_idx=li)


def read_bal(path, native=None) -> BAProblem:
    """native=None: the C++ reader when libgbp_hip.so is built, this Python reader otherwise (pure host I/O either way)."""
    if native is None:
        from . import _capi
        native = _capi.available()
    if native:
        return read_bal_native(path)
    with open(path, 'r') as f:
        lines = f.read().split('\n')
    pos = 0
    while True:
        if pos >= len(lines):
            raise ValueError(f"{path}: no header line found")
        tok = lines[pos].split()
        pos += 1
        if tok and tok[0] != '#':
            break
    n_cams, n_lmks, n_obs = (int(x) for x in tok[:3])
    K = np.array([float(x) for x in lines[pos].split()[:4]], dtype=np.float64)
    pos += 1
    obs = np.array([ln.split()[:4] for ln in lines[pos:pos + n_obs]], dtype=np.float64)
    if obs.shape != (n_obs, 4):
        raise ValueError(f"{path}: expected {n_obs} observation rows")
    pos += n_obs
    n_scalars = 6 * n_cams + 3 * n_lmks
    vals = np.array([ln.split()[0] for ln in lines[pos:pos + n_scalars]], dtype=np.float64)
    if vals.shape[0] != n_scalars:
        raise ValueError(f"{path}: expected {n_scalars} initialisation scalars")
    return BAProblem(K=K,
                     cam_means=vals[:6 * n_cams].reshape(n_cams, 6).copy(),
                     lmk_means=vals[6 * n_cams:].reshape(n_lmks, 3).copy(),
                     meas=np.ascontiguousarray(obs[:, 2:4]),
                     cam_idx=obs[:, 0].astype(np.int32),
                     lmk_idx=obs[:, 1].astype(np.int32))


def reference_factor_order(cam_idx) -> np.ndarray:
    """Permutation file-order -> reference factor order.

    create_ba_graph scans the observations once per camera (gbp/gbp_ba.py:128-130), so factor ids
    are camera-major with file order preserved inside a camera: a stable sort by camera id.
    """
    return np.argsort(np.asarray(cam_idx), kind='stable')
