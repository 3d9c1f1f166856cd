"""Landmark-sharded GBP sweep over several GPUs of one node.

The reference is single-process; this is the part of the engine that has no counterpart there.
One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI):

  * landmarks are cut into `world` contiguous ranges balanced by factor count; a factor lives with
    its landmark, so robustify / relinearise / messages and the landmark beliefs are rank-local;
  * cameras are replicated.  Per sweep every rank produces its camera partial sums (C x 27 packed
    doubles, no prior), ONE all-gather moves them (108 KB per rank at C = 500), and every rank adds
    them in rank order + prior and solves the 6x6 -- bitwise identical camera beliefs on all ranks;
  * diagnostics (ARE / energy) need one 2-double all-reduce; generate_priors_var needs one MAX
    all-reduce over the per-camera factor maxima at set-up.

The HIP work goes through the C ABI (gbp_ba_shard_begin / gbp_ba_shard_end); torch only provides the
exchange buffers, the stream and the collective.  `engine_factory` lets the CPU tests run the very
same host logic over gloo with a test double for the engine.
"""
from __future__ import annotations

import numpy as np

from .synthetic import BAProblem


def partition_landmarks(lmk_idx, n_lmks, world):
    """Contiguous landmark ranges [b[r], b[r+1]) with (nearly) equal factor counts."""
    deg = np.bincount(np.asarray(lmk_idx), minlength=n_lmks).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(deg)])
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(cum, target, side='left'))
        bounds.append(min(max(b, bounds[-1]), n_lmks))
    bounds.append(n_lmks)
    return np.array(bounds, dtype=np.int64)


def local_problem(problem: BAProblem, lo: int, hi: int) -> BAProblem:
    """The sub-problem of landmarks [lo, hi): their factors (file order kept), all cameras."""
    sel = (problem.lmk_idx >= lo) & (problem.lmk_idx < hi)
    return BAProblem(K=problem.K, cam_means=problem.cam_means, lmk_means=problem.lmk_means[lo:hi],
                     meas=problem.meas[sel], cam_idx=problem.cam_idx[sel],
                     lmk_idx=(problem.lmk_idx[sel] - lo).astype(np.int32))


class _HipShard:
    """One rank's BAEngine + its exchange buffers on the GPU."""

    def __init__(self, problem, device, fused, **cfg):
        import torch
        from .engine import BAEngine
        from ._capi import CAM_PARTIAL_DOUBLES
        self.torch = torch
        self.engine = BAEngine.from_problem(problem, device=device, fused=fused, **cfg)
        self.partial_doubles = CAM_PARTIAL_DOUBLES
        self.device = torch.device('cuda', device)
        # kernels and the collective are ordered on ONE side stream of torch's (a real handle: the legacy default
        # stream is the null pointer, which gbp_ba_set_stream reads as "the engine's own stream")
        self.stream = torch.cuda.Stream(self.device)
        self.engine.set_stream(self.stream.cuda_stream)

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def new_buffer(self, n):
        return self.torch.empty(n, dtype=self.torch.float64, device=self.device)

    def begin(self, partial, with_messages, robustify, local_relin):
        self.engine.shard_begin(partial.data_ptr(), with_messages, robustify, local_relin)

    def end(self, gathered, world):
        self.engine.shard_end(gathered.data_ptr(), world)

    def to_tensor(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a), device=self.device)

    def sync(self):
        self.stream.synchronize()


class ShardedBA:
    """BAFactorGraph surface (gbp_ba.py:12-69) over `world` ranks; every rank calls every method."""

    def __init__(self, problem: BAProblem, device=0, fused=True, engine_factory=None, dist=None, **cfg):
        if dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.C, self.F_total, self.L_total = problem.n_cams, problem.n_factors, problem.n_lmks
        self.bounds = partition_landmarks(problem.lmk_idx, problem.n_lmks, self.world)
        lo, hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        self.lmk_range = (lo, hi)
        local = local_problem(problem, lo, hi)
        factory = engine_factory or (lambda p: _HipShard(p, device, fused, **cfg))
        self.shard = factory(local)
        self.engine = self.shard.engine
        self.F, self.L = local.n_factors, local.n_lmks
        n = self.C * self.shard.partial_doubles
        self._partial = self.shard.new_buffer(n)
        self._gathered = self.shard.new_buffer(n * self.world)

    # ---- set-up ---------------------------------------------------------------------------
    def generate_priors_var(self, weaker_factor=100.0):
        cam_max, lmk_max = self.engine.factor_lambda_max()
        with self._ctx():
            t = self.shard.to_tensor(cam_max)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)      # max over ALL factors of a camera (gbp_ba.py:28-31)
        self.shard.sync()
        w2 = float(weaker_factor) ** 2
        self.engine.set_prior_scalars(t.cpu().numpy() / w2, lmk_max / w2)

    def weaken_priors(self, f):
        self.engine.weaken_priors(f)

    # ---- sweep -----------------------------------------------------------------------------
    def _ctx(self):
        import contextlib
        return self.shard.stream_ctx() if hasattr(self.shard, 'stream_ctx') else contextlib.nullcontext()

    def _exchange(self):
        with self._ctx():
            self.dist.all_gather_into_tensor(self._gathered, self._partial)

    def update_all_beliefs(self):
        self.shard.begin(self._partial, False, False, False)
        self._exchange()
        self.shard.end(self._gathered, self.world)

    def synchronous_iteration(self, local_relin=True, robustify=False):
        self.shard.begin(self._partial, True, robustify, local_relin)
        self._exchange()
        self.shard.end(self._gathered, self.world)

    def iterate(self, n, robustify=True, local_relin=True):
        for _ in range(int(n)):
            self.synchronous_iteration(local_relin=local_relin, robustify=robustify)

    def set_iters_since_relin(self, v):
        if not np.isscalar(v):
            raise ValueError("sharded graphs take a scalar iters_since_relin (ba.py:91-93 sets all factors alike)")
        self.engine.set_iters_since_relin(v)

    # ---- diagnostics -----------------------------------------------------------------------
    def _residual_sums(self):
        sums = self.engine.residual_sums()
        with self._ctx():
            t = self.shard.to_tensor(sums)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self.shard.sync()
        return t.cpu().numpy()

    def are(self):
        return float(self._residual_sums()[0]) / self.F_total         # / len(self.factors)  gbp_ba.py:69

    def energy(self):
        return float(self._residual_sums()[1])

    # ---- views -----------------------------------------------------------------------------
    def camera_beliefs(self):
        ce, cl, _, _ = self.engine.beliefs()
        return ce, cl

    def local_landmark_beliefs(self):
        _, _, le, ll = self.engine.beliefs()
        return self.lmk_range, le, ll

    def sync(self):
        self.shard.sync()

    def set_kernel_timing(self, on):
        self.engine.set_kernel_timing(on)

    def kernel_timing(self):
        return self.engine.kernel_timing()

    def info(self):
        return self.engine.info()
