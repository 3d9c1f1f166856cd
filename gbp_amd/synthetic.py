"""Synthetic bundle-adjustment problems in the reference's BAL-style layout.

The reference ships only five small TUM-derived problems (data/README.md:5-16 describes the
text layout).  BASELINE.json's headline workload (500 cameras x 100k landmarks x 1M reprojection
factors) does not exist as a file, so this module generates it deterministically
(SURVEY.md section 8d spec): same intrinsics as data/fr1desk.txt:9, cameras on a shell looking
at a ball of landmarks, every landmark seen by exactly `obs_per_lmk` distinct cameras,
measurements = exact pin-hole projection + 1 px Gaussian noise, camera-major factor order.

Everything is vectorised numpy; the 1M-factor problem takes a few seconds.
"""
from __future__ import annotations

import dataclasses

import numpy as np

FR1DESK_K = (517.306408, 516.469215, 318.64304, 255.313989)  # fx fy cx cy, data/fr1desk.txt:9


@dataclasses.dataclass
class BAProblem:
    """Arrays of one BA problem, factor order = file order (camera-major for our generator)."""
    K: np.ndarray          # (4,)  fx fy cx cy
    cam_means: np.ndarray  # (C,6) t(3), axis-angle(3) of T_cw
    lmk_means: np.ndarray  # (L,3)
    meas: np.ndarray       # (F,2) pixels
    cam_idx: np.ndarray    # (F,) int32
    lmk_idx: np.ndarray    # (F,) int32

    @property
    def n_cams(self):
        return int(self.cam_means.shape[0])

    @property
    def n_lmks(self):
        return int(self.lmk_means.shape[0])

    @property
    def n_factors(self):
        return int(self.meas.shape[0])


def rodrigues(w):
    """Batched axis-angle -> rotation matrix, (N,3) -> (N,3,3)."""
    w = np.atleast_2d(np.asarray(w, dtype=np.float64))
    th = np.linalg.norm(w, axis=1)
    safe = np.where(th > 0, th, 1.0)
    a = np.where(th > 0, np.sin(th) / safe, 1.0)
    b = np.where(th > 0, (1.0 - np.cos(th)) / (safe * safe), 0.5)
    Wh = np.zeros((w.shape[0], 3, 3))
    Wh[:, 0, 1], Wh[:, 0, 2] = -w[:, 2], w[:, 1]
    Wh[:, 1, 0], Wh[:, 1, 2] = w[:, 2], -w[:, 0]
    Wh[:, 2, 0], Wh[:, 2, 1] = -w[:, 1], w[:, 0]
    return np.eye(3)[None] + a[:, None, None] * Wh + b[:, None, None] * (Wh @ Wh)


def _log_so3(R):
    """Batched rotation matrix -> axis-angle for angles away from 0 and pi."""
    tr = np.clip((np.trace(R, axis1=1, axis2=2) - 1.0) * 0.5, -1.0, 1.0)
    th = np.arccos(tr)
    v = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
    return v * (th / (2.0 * np.sin(th)))[:, None]


def _look_at_cameras(rng, n, target_jitter, r_lo, r_hi):
    """n world->camera poses on a shell, optical axis through (origin + jitter), random roll."""
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos = d * rng.uniform(r_lo, r_hi, size=(n, 1))
    tgt = rng.uniform(-target_jitter, target_jitter, size=(n, 3))
    fwd = tgt - pos
    fwd /= np.linalg.norm(fwd, axis=1, keepdims=True)
    helper = np.where(np.abs(fwd[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    right = np.cross(helper, fwd)
    right /= np.linalg.norm(right, axis=1, keepdims=True)
    down = np.cross(fwd, right)
    roll = rng.uniform(0.0, 2.0 * np.pi, size=n)
    cr, sr = np.cos(roll)[:, None], np.sin(roll)[:, None]
    x_ax = cr * right + sr * down
    y_ax = -sr * right + cr * down
    R_cw = np.stack([x_ax, y_ax, fwd], axis=1)  # rows = camera axes in world coordinates
    t = -np.einsum('nij,nj->ni', R_cw, pos)
    return R_cw, t


def make_synthetic(n_cams=500, n_lmks=100_000, obs_per_lmk=10, seed=0, K=FR1DESK_K,
                   width=640.0, height=480.0, pix_noise=1.0, cam_t_noise=0.02,
                   ball_radius=2.0, shell=(5.0, 6.0), target_jitter=0.2, min_depth=0.5, window=None, closures=0.0, lmk_seed=None):
    """Generate a BA problem with exactly n_lmks*obs_per_lmk factors (camera-major order).

    window = w: a SEQUENCE instead of the headline graph's all-see-all -- every landmark is seen by obs_per_lmk cameras out of w
    consecutive ones, and the landmarks are numbered along the trajectory (by the centre of their window), the way a SLAM front end
    or an incremental reconstruction numbers them (the reference's fr1desk files: a landmark's cameras span 2 .. 46 consecutive
    keyframes).  closures = p: that fraction of the landmarks is seen from anywhere along the trajectory instead (places visited
    again).  The default (None) is the graph of BASELINE configs 4-5 and draws exactly the random numbers it always drew.
    lmk_seed = k: the CAMERAS (poses and their initial estimates) are those of `seed`, whatever k; the landmarks, who sees them and the
    pixel noise come from a generator of their own, (seed, k) -- so that every rank of a sharded job can make its own landmarks against
    the cameras all ranks share (bench.py's secondary workload: ShardedBA(local_shard=True))."""
    if obs_per_lmk > n_cams:
        raise ValueError("obs_per_lmk cannot exceed n_cams")
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K

    # cameras: rejection-sample rotations whose axis-angle is well inside (0.05, 3.0)
    R_list, t_list = [], []
    have = 0
    while have < n_cams:
        R, t = _look_at_cameras(rng, 2 * (n_cams - have) + 8, target_jitter, *shell)
        w = _log_so3(R)
        th = np.linalg.norm(w, axis=1)
        ok = (th > 0.05) & (th < 3.0)
        R_list.append(R[ok]); t_list.append(t[ok]); have += int(ok.sum())
    R_cw = np.concatenate(R_list)[:n_cams]
    t_cw = np.concatenate(t_list)[:n_cams]
    w_cw = _log_so3(R_cw)
    R_cw = rodrigues(w_cw)  # the rotation the engine will actually reconstruct
    t_init_shared = None
    if lmk_seed is not None:
        t_init_shared = t_cw + rng.normal(scale=cam_t_noise, size=t_cw.shape)      # the cameras' estimates must not depend on k
        rng = np.random.default_rng([int(seed), int(lmk_seed)])

    # landmarks + visibility, chunked so that memory stays bounded at 100k x 500
    if window is not None and not obs_per_lmk <= window <= n_cams:
        raise ValueError("window must lie between obs_per_lmk and n_cams")
    lmk = np.empty((n_lmks, 3))
    chosen = np.empty((n_lmks, obs_per_lmk), dtype=np.int32)
    centre = np.empty(n_lmks)
    done = 0
    chunk = 8192
    stalls = 0
    while done < n_lmks:
        m = min(chunk, n_lmks - done)
        v = rng.normal(size=(m, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        pts = v * (ball_radius * rng.uniform(size=(m, 1)) ** (1.0 / 3.0))
        pc = np.einsum('cij,mj->mci', R_cw, pts) + t_cw[None]
        z = pc[..., 2]
        zs = np.where(z > min_depth, z, 1.0)
        u = fx * pc[..., 0] / zs + cx
        vv = fy * pc[..., 1] / zs + cy
        vis = (z > min_depth) & (u >= 0) & (u < width) & (vv >= 0) & (vv < height)
        if window is not None:
            ctr = rng.uniform(window / 2.0, n_cams - window / 2.0, size=m)
            near = np.abs(np.arange(n_cams)[None, :] + 0.5 - ctr[:, None]) <= window / 2.0
            if closures > 0.0:
                near |= (rng.uniform(size=m) < closures)[:, None]
            vis &= near
        keys = np.where(vis, rng.uniform(size=vis.shape), 2.0)
        pick = np.argpartition(keys, obs_per_lmk - 1, axis=1)[:, :obs_per_lmk]
        good = np.take_along_axis(keys, pick, axis=1).max(axis=1) < 1.5
        k = int(good.sum())
        k = min(k, n_lmks - done)
        stalls = stalls + 1 if k == 0 else 0
        if stalls > 50:
            raise ValueError(f"no landmark is visible from {obs_per_lmk} of the {n_cams} cameras")
        sel = np.nonzero(good)[0][:k]
        lmk[done:done + k] = pts[sel]
        chosen[done:done + k] = np.sort(pick[sel], axis=1)
        if window is not None:
            centre[done:done + k] = ctr[sel]
        done += k
    if window is not None:                                   # numbered along the trajectory
        along = np.argsort(centre, kind='stable')
        lmk, chosen = lmk[along], chosen[along]

    cam_idx = chosen.reshape(-1)
    lmk_idx = np.repeat(np.arange(n_lmks, dtype=np.int32), obs_per_lmk)
    order = np.argsort(cam_idx, kind='stable')  # camera-major, landmarks ascending inside a camera
    cam_idx = cam_idx[order].astype(np.int32)
    lmk_idx = lmk_idx[order].astype(np.int32)

    pc = np.einsum('fij,fj->fi', R_cw[cam_idx], lmk[lmk_idx]) + t_cw[cam_idx]
    depth = pc[:, 2]
    meas = np.stack([fx * pc[:, 0] / depth + cx, fy * pc[:, 1] / depth + cy], axis=1)
    meas = meas + rng.normal(scale=pix_noise, size=meas.shape)

    # initial estimates: noisy camera translations, exact rotations (data/fr1desk.txt:4-7)
    t_init = t_init_shared if t_init_shared is not None else t_cw + rng.normal(scale=cam_t_noise, size=t_cw.shape)
    cam_means = np.concatenate([t_init, w_cw], axis=1)

    # landmarks start on the ray of their first observation at the mean scene depth
    first = np.full(n_lmks, -1, dtype=np.int64)
    fidx = np.arange(cam_idx.shape[0])
    # first occurrence in file order == observation with the lowest camera id
    rev = fidx[::-1]
    first[lmk_idx[rev]] = rev
    mean_depth = float(depth.mean())
    uv = meas[first]
    ray = np.stack([(uv[:, 0] - cx) / fx, (uv[:, 1] - cy) / fy, np.ones(n_lmks)], axis=1) * mean_depth
    c0 = cam_idx[first]
    lmk_init = np.einsum('nji,nj->ni', R_cw[c0], ray - t_init[c0])  # R^T (p_c - t)

    return BAProblem(K=np.asarray(K, dtype=np.float64), cam_means=cam_means, lmk_means=lmk_init,
                     meas=meas, cam_idx=cam_idx, lmk_idx=lmk_idx)


def write_bal(problem: BAProblem, path, header="synthetic"):
    """Write `problem` in the text layout the reference parses (utils/read_balfile.py:4-37)."""
    with open(path, 'w') as f:
        f.write(f"# Dataset: {header}\n\n")
        f.write(f"{problem.n_cams} {problem.n_lmks} {problem.n_factors}\n")
        f.write("%.9g %.9g %.9g %.9g\n" % tuple(problem.K))
        for c, l, (u, v) in zip(problem.cam_idx, problem.lmk_idx, problem.meas):
            f.write("%d %d     %.17e %.17e\n" % (c, l, u, v))
        for val in problem.cam_means.reshape(-1):
            f.write("%.17e\n" % val)
        for val in problem.lmk_means.reshape(-1):
            f.write("%.17e\n" % val)
