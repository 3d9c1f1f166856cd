"""BAEngine: the GBP bundle-adjustment sweep on one MI355X, behind the C ABI of include/gbp_ba.h.

Method names follow the reference's BAFactorGraph / FactorGraph (gbp/gbp_ba.py:12-69,
gbp/gbp.py:36-92) so that callers and parity tests read like the reference.  All compute happens
in libgbp_hip.so on the GPU; constructing an engine without a gfx950 device raises GbpError.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _capi
from ._capi import check, dptr, iptr, bptr, f64, i32


def eval_fn(K4, x9, device=0):
    """meas_fn / jac_fn of the reprojection factor on the device (gbp_ba_eval_fn): (h (n,2), J (n,2,9), h_proj (n,2))."""
    lib = _capi.load()
    K = f64(np.asarray(K4, dtype=np.float64).reshape(-1), (4,))
    x = f64(x9).reshape(-1, 9)
    n = x.shape[0]
    h, J, hp = np.empty((n, 2)), np.empty((n, 2, 9)), np.empty((n, 2))
    check(lib.gbp_ba_eval_fn(dptr(K), n, dptr(x), dptr(h), dptr(J), dptr(hp), int(device)))
    return h, J, hp


def _sweep_flags(fused):
    return 0 if fused is None else (_capi.FLAG_FORCE_FUSED if fused else _capi.FLAG_NO_FUSED)


class BAEngine:
    def __init__(self, K, cam_means, lmk_means, meas, cam_idx, lmk_idx, *, gauss_noise_std=2.0, loss=None,
                 Nstds=3.0, beta=0.01, num_undamped_iters=6, min_linear_iters=8, eta_damping=0.4,
                 device=0, fused=None, device_pointers=None):
        """fused: None = the library picks the sweep (the fused one -- with per-workgroup camera windows where each workgroup's tiles meet
        few of the cameras -- unless neither all cameras nor the workgroups' camera sets fit one LDS table; plan_info() says what it chose);
        True = the fused sweep whatever the sparseness rule says (GBP_FLAG_FORCE_FUSED); False = the general sweep (GBP_FLAG_NO_FUSED).
        device_pointers = (C, L, F): the five arrays are then integer DEVICE addresses on `device` (float64 cam_means[C,6],
        lmk_means[L,3], meas[F,2]; int32 cam_idx[F], lmk_idx[F]) and nothing is uploaded (GBP_FLAG_DEVICE_INPUT)."""
        self._lib = _capi.load()
        if device_pointers is not None:
            return self._init_from_device(K, cam_means, lmk_means, meas, cam_idx, lmk_idx, device_pointers,
                                          gauss_noise_std, loss, Nstds, beta, num_undamped_iters, min_linear_iters,
                                          eta_damping, device, fused)
        K = np.asarray(K, dtype=np.float64)
        if K.shape == (3, 3):
            K = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
        K = f64(K.reshape(-1), (4,))
        cam_means = f64(cam_means).reshape(-1, 6)
        lmk_means = f64(lmk_means).reshape(-1, 3)
        meas = f64(meas).reshape(-1, 2)
        cam_idx = i32(cam_idx).reshape(-1)
        lmk_idx = i32(lmk_idx).reshape(-1)
        self.C, self.L, self.F = cam_means.shape[0], lmk_means.shape[0], meas.shape[0]
        if cam_idx.shape[0] != self.F or lmk_idx.shape[0] != self.F:
            raise ValueError("cam_idx / lmk_idx / meas length mismatch")
        if loss not in _capi.LOSS:
            raise ValueError(f"unknown loss {loss!r} (None, 'huber', 'constant')")
        self.K = K.copy()
        self.eta_damping = float(eta_damping)
        d = _capi.Desc()
        d.n_cams, d.n_lmks, d.n_factors, d.device = self.C, self.L, self.F, int(device)
        d.K[:] = list(K)
        d.cam_means, d.lmk_means, d.meas = dptr(cam_means), dptr(lmk_means), dptr(meas)
        d.cam_idx, d.lmk_idx = iptr(cam_idx), iptr(lmk_idx)
        d.gauss_noise_std = float(gauss_noise_std)
        d.loss = _capi.LOSS[loss]
        d.num_undamped_iters, d.min_linear_iters = int(num_undamped_iters), int(min_linear_iters)
        d.flags = _sweep_flags(fused)
        d.nstds, d.beta, d.eta_damping = float(Nstds), float(beta), float(eta_damping)
        self._h = ct.c_void_p()
        check(self._lib.gbp_ba_create(ct.byref(self._h), ct.byref(d)))

    def _init_from_device(self, K, cam_means, lmk_means, meas, cam_idx, lmk_idx, sizes, gauss_noise_std, loss, Nstds, beta,
                          num_undamped_iters, min_linear_iters, eta_damping, device, fused):
        K = np.asarray(K, dtype=np.float64)
        if K.shape == (3, 3):                                # the host constructor takes either form too
            K = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
        K = f64(K.reshape(-1), (4,))
        self.C, self.L, self.F = (int(v) for v in sizes)
        if loss not in _capi.LOSS:
            raise ValueError(f"unknown loss {loss!r} (None, 'huber', 'constant')")
        self.K = K.copy()
        self.eta_damping = float(eta_damping)
        d = _capi.Desc()
        d.n_cams, d.n_lmks, d.n_factors, d.device = self.C, self.L, self.F, int(device)
        d.K[:] = list(K)
        d.cam_means = ct.cast(ct.c_void_p(int(cam_means)), _capi._dp)
        d.lmk_means = ct.cast(ct.c_void_p(int(lmk_means)), _capi._dp)
        d.meas = ct.cast(ct.c_void_p(int(meas)), _capi._dp)
        d.cam_idx = ct.cast(ct.c_void_p(int(cam_idx)), _capi._ip)
        d.lmk_idx = ct.cast(ct.c_void_p(int(lmk_idx)), _capi._ip)
        d.gauss_noise_std = float(gauss_noise_std)
        d.loss = _capi.LOSS[loss]
        d.num_undamped_iters, d.min_linear_iters = int(num_undamped_iters), int(min_linear_iters)
        d.flags = _sweep_flags(fused) | _capi.FLAG_DEVICE_INPUT
        d.nstds, d.beta, d.eta_damping = float(Nstds), float(beta), float(eta_damping)
        self._h = ct.c_void_p()
        check(self._lib.gbp_ba_create(ct.byref(self._h), ct.byref(d)))

    @classmethod
    def from_problem(cls, p, **kw):
        return cls(p.K, p.cam_means, p.lmk_means, p.meas, p.cam_idx, p.lmk_idx, **kw)

    def close(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._lib.gbp_ba_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- priors (gbp_ba.py:20-52) ---------------------------------------------------------
    def generate_priors_var(self, weaker_factor=100.0):
        check(self._lib.gbp_ba_generate_priors(self._h, float(weaker_factor)))

    def factor_lambda_max(self):
        cm, lm = np.empty(self.C), np.empty(self.L)
        check(self._lib.gbp_ba_factor_lambda_max(self._h, dptr(cm), dptr(lm)))
        return cm, lm

    def set_prior_scalars(self, cam_lambda, lmk_lambda):
        a, b = f64(cam_lambda, (self.C,)), f64(lmk_lambda, (self.L,))
        check(self._lib.gbp_ba_set_prior_scalars(self._h, dptr(a), dptr(b)))

    def set_priors(self, cam_eta, cam_lam, lmk_eta, lmk_lam):
        a, b = f64(cam_eta, (self.C, 6)), f64(cam_lam, (self.C, 6, 6))
        c, d = f64(lmk_eta, (self.L, 3)), f64(lmk_lam, (self.L, 3, 3))
        check(self._lib.gbp_ba_set_priors(self._h, dptr(a), dptr(b), dptr(c), dptr(d)))

    def set_priors_var(self, priors):
        """priors: covariance per variable, cameras then landmarks (gbp_ba.py:44-52)."""
        if len(priors) != self.C + self.L:
            raise ValueError("need one covariance per variable node")
        cm, lm = self.means()
        cl = np.array([np.linalg.inv(np.asarray(priors[v], dtype=np.float64)) for v in range(self.C)]).reshape(self.C, 6, 6)
        ll = np.array([np.linalg.inv(np.asarray(priors[self.C + v], dtype=np.float64)) for v in range(self.L)]).reshape(self.L, 3, 3)
        self.set_priors(np.einsum('nij,nj->ni', cl, cm), cl, np.einsum('nij,nj->ni', ll, lm), ll)

    def weaken_priors(self, weakening_factor):
        check(self._lib.gbp_ba_weaken_priors(self._h, float(weakening_factor)))

    # ---- the sweep (gbp.py:56-92) ----------------------------------------------------------
    def update_all_beliefs(self):
        check(self._lib.gbp_ba_update_beliefs(self._h))

    def synchronous_iteration(self, local_relin=True, robustify=False):
        check(self._lib.gbp_ba_iterate(self._h, 1, int(bool(robustify)), int(bool(local_relin))))

    def iterate(self, n, robustify=True, local_relin=True):
        check(self._lib.gbp_ba_iterate(self._h, int(n), int(bool(robustify)), int(bool(local_relin))))

    # the same sweep stage by stage (gbp.py:46-84); synchronous_iteration is the fused form of these four in a row
    def robustify_all_factors(self):
        check(self._lib.gbp_ba_robustify(self._h))

    def relinearise_factors(self):
        check(self._lib.gbp_ba_relinearise(self._h))

    def compute_all_messages(self, local_relin=True):
        check(self._lib.gbp_ba_compute_messages(self._h, int(bool(local_relin))))

    def compute_all_factors(self):
        check(self._lib.gbp_ba_compute_factors(self._h))

    def sync(self):
        check(self._lib.gbp_ba_sync(self._h))

    def set_stream(self, stream_ptr):
        check(self._lib.gbp_ba_set_stream(self._h, ct.c_void_p(stream_ptr) if stream_ptr else None))

    # ---- sharded sweep ---------------------------------------------------------
    def shard_begin(self, partial_ptr, with_messages=True, robustify=True, local_relin=True):
        check(self._lib.gbp_ba_shard_begin(self._h, int(with_messages), int(robustify), int(local_relin),
                                           ct.c_void_p(partial_ptr)))

    def shard_end(self, gathered_ptr, n_ranks):
        check(self._lib.gbp_ba_shard_end(self._h, ct.c_void_p(gathered_ptr), int(n_ranks)))

    # ---- in-library sharded loop (include/gbp_ba.h: gbp_ba_iterate_sharded) ---------------------------
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id made on one rank; every rank of the job passes the same bytes to comm_init_rccl."""
        lib = _capi.load()
        buf = ct.create_string_buffer(_capi.COMM_ID_BYTES)
        check(lib.gbp_ba_comm_unique_id(buf, _capi.torch_rccl_path()))
        return buf.raw

    def comm_init_rccl(self, unique_id, rank, n_ranks, always_exchange=False):
        if len(unique_id) != _capi.COMM_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        check(self._lib.gbp_ba_comm_init_rccl(self._h, ct.c_char_p(bytes(unique_id)), int(rank), int(n_ranks),
                                              _capi.XCH_ALWAYS if always_exchange else 0, _capi.torch_rccl_path()))

    def set_exchange(self, fn, rank, n_ranks, always_exchange=False):
        """fn(send_ptr, recv_ptr, count, stream_ptr) -> 0: a Python all-gather of the camera partial sums (tests, MPI)."""
        if fn is None:
            self._xch_cb = None
            check(self._lib.gbp_ba_set_exchange(self._h, None, None, int(rank), int(n_ranks), 0))
            return
        def tramp(ctx, send, recv, count, stream):
            try:
                return int(fn(send, recv, int(count), stream) or 0)
            except Exception:                      # nothing may propagate through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._xch_cb = _capi.EXCHANGE_FN(tramp)      # kept alive with the engine
        check(self._lib.gbp_ba_set_exchange(self._h, ct.cast(self._xch_cb, ct.c_void_p), None, int(rank), int(n_ranks),
                                            _capi.XCH_ALWAYS if always_exchange else 0))

    def peer_export(self, n_ranks, same_process=False):
        """This rank's mailbox for the peer-store exchange (gbp_ba_peer_export): 64 bytes to hand to every other rank."""
        buf = ct.create_string_buffer(_capi.PEER_HANDLE_BYTES)
        check(self._lib.gbp_ba_peer_export(self._h, int(n_ranks), buf, _capi.PEER_SAME_PROCESS if same_process else 0))
        return buf.raw

    def peer_connect(self, rank, handles, same_process=False, rendezvous=False):
        """handles: the peer_export() bytes of all ranks in rank order.  From here on iterate_sharded / update_beliefs_sharded
        exchange the camera partial sums by direct stores into the ranks' mailboxes (no collective call)."""
        blob = b''.join(bytes(x) for x in handles)
        if len(blob) != _capi.PEER_HANDLE_BYTES * len(handles):
            raise ValueError("every handle must be the 64 bytes of peer_export()")
        flags = (_capi.PEER_SAME_PROCESS if same_process else 0) | (_capi.PEER_RENDEZVOUS if rendezvous else 0)
        check(self._lib.gbp_ba_peer_connect(self._h, int(rank), len(handles), ct.c_char_p(blob), flags))

    def peer_selftest(self, timeout_ms=5000):
        """After peer_connect and a barrier of the side channel, on every rank: a tagged probe row to every rank's mailbox and the check
        of everybody's on arrival (gbp_ba_peer_selftest).  Raises GbpError naming the pair that failed."""
        check(self._lib.gbp_ba_peer_selftest(self._h, int(timeout_ms)))

    def comm_destroy(self):
        check(self._lib.gbp_ba_comm_destroy(self._h))

    def iterate_sharded(self, n, robustify=True, local_relin=True):
        check(self._lib.gbp_ba_iterate_sharded(self._h, int(n), int(bool(robustify)), int(bool(local_relin))))

    def update_beliefs_sharded(self):
        check(self._lib.gbp_ba_update_beliefs_sharded(self._h))

    # ---- diagnostics (gbp_ba.py:61-69, gbp.py:36-44) ---------------------------------------
    def are(self):
        v = ct.c_double()
        check(self._lib.gbp_ba_are(self._h, ct.byref(v)))
        return v.value

    def energy(self):
        v = ct.c_double()
        check(self._lib.gbp_ba_energy(self._h, ct.byref(v)))
        return v.value

    def residual_sums(self):
        out = np.empty(2)
        check(self._lib.gbp_ba_residual_sums(self._h, dptr(out)))
        return out

    # ---- views -----------------------------------------------------------------------------
    def _four(self, fn):
        ce, cl = np.empty((self.C, 6)), np.empty((self.C, 6, 6))
        le, ll = np.empty((self.L, 3)), np.empty((self.L, 3, 3))
        check(fn(self._h, dptr(ce), dptr(cl), dptr(le), dptr(ll)))
        return ce, cl, le, ll

    def beliefs(self):
        return self._four(self._lib.gbp_ba_get_beliefs)

    def priors(self):
        return self._four(self._lib.gbp_ba_get_priors)

    def means(self):
        cm, lm = np.empty((self.C, 6)), np.empty((self.L, 3))
        check(self._lib.gbp_ba_get_means(self._h, dptr(cm), dptr(lm)))
        return cm, lm

    def covariances(self):
        cs, ls = np.empty((self.C, 6, 6)), np.empty((self.L, 3, 3))
        check(self._lib.gbp_ba_get_covariances(self._h, dptr(cs), dptr(ls)))
        return cs, ls

    def messages(self, f0=0, n=None):
        n = self.F - f0 if n is None else n
        ce, cl = np.empty((n, 6)), np.empty((n, 6, 6))
        le, ll = np.empty((n, 3)), np.empty((n, 3, 3))
        check(self._lib.gbp_ba_get_messages(self._h, int(f0), int(n), dptr(ce), dptr(cl), dptr(le), dptr(ll)))
        return ce, cl, le, ll

    def factors(self, f0=0, n=None, dense=True):
        n = self.F - f0 if n is None else n
        eta = np.empty((n, 9)) if dense else None
        lam = np.empty((n, 9, 9)) if dense else None
        lp, cam, lmk, z = np.empty((n, 9)), np.empty(n, np.int32), np.empty(n, np.int32), np.empty((n, 2))
        check(self._lib.gbp_ba_get_factors(self._h, int(f0), int(n), dptr(eta), dptr(lam), dptr(lp), iptr(cam), iptr(lmk), dptr(z)))
        return dict(eta=eta, lam=lam, linpoint=lp, cam=cam, lmk=lmk, z=z)

    def relin_state(self):
        it, d = np.empty(self.F, np.int32), np.empty(self.F)
        av, rb = np.empty(self.F), np.empty(self.F, np.uint8)
        check(self._lib.gbp_ba_get_relin_state(self._h, iptr(it), dptr(d), dptr(av), bptr(rb)))
        return dict(iters_since_relin=it, eta_damping=d, adaptive_var=av, robust_flag=rb)

    def iters_since_relin(self):
        """iters_since_relin of every factor alone (4 bytes per factor instead of the 21 of relin_state)."""
        it = np.empty(self.F, np.int32)
        check(self._lib.gbp_ba_get_relin_state(self._h, iptr(it), None, None, None))
        return it

    def relin_state_range(self, f0, n):
        it, d = np.empty(n, np.int32), np.empty(n)
        av, rb = np.empty(n), np.empty(n, np.uint8)
        check(self._lib.gbp_ba_get_relin_state_range(self._h, int(f0), int(n), iptr(it), dptr(d), dptr(av), bptr(rb)))
        return dict(iters_since_relin=it, eta_damping=d, adaptive_var=av, robust_flag=rb)

    def count_relinearising(self):
        """Number of factors with iters_since_relin == 0 (the loop of ba.py:96-99), reduced on the device."""
        v = ct.c_int64()
        check(self._lib.gbp_ba_count_relinearising(self._h, ct.byref(v)))
        return v.value

    def relin_counts(self, n):
        """Factors that relinearised in each of the last n sweeps (oldest first)."""
        out = np.empty(int(n), np.int32)
        check(self._lib.gbp_ba_get_relin_counts(self._h, iptr(out), int(n)))
        return out

    def set_iters_since_relin(self, v):
        if np.isscalar(v):
            check(self._lib.gbp_ba_fill_iters_since_relin(self._h, int(v)))
        else:
            a = i32(v, (self.F,))
            check(self._lib.gbp_ba_set_iters_since_relin(self._h, iptr(a)))

    # ---- instrumentation -------------------------------------------------------------------
    # ---- streaming means for a viewer (include/gbp_ba.h: gbp_ba_means_snapshot) ---------------------
    def means_snapshot(self):
        check(self._lib.gbp_ba_means_snapshot(self._h))

    def means_fetch(self, wait=True):
        cm, lm = np.empty((self.C, 6)), np.empty((self.L, 3))
        check(self._lib.gbp_ba_means_fetch(self._h, dptr(cm), dptr(lm), int(bool(wait))))
        return cm, lm

    # ---- checkpoint / restore (include/gbp_ba.h: gbp_ba_save_state) --------------------------------
    def save_state(self):
        """Everything a sweep reads or writes, as one uint8 array (restores only into an engine of the same graph)."""
        n = ct.c_uint64()
        check(self._lib.gbp_ba_state_size(self._h, ct.byref(n)))
        buf = np.empty(n.value, dtype=np.uint8)
        check(self._lib.gbp_ba_save_state(self._h, buf.ctypes.data_as(ct.c_void_p), n))
        return buf

    def load_state(self, blob):
        buf = np.ascontiguousarray(blob, dtype=np.uint8)
        check(self._lib.gbp_ba_load_state(self._h, buf.ctypes.data_as(ct.c_void_p), ct.c_uint64(buf.size)))

    def snapshot_state(self):
        """Checkpoint kept on the device (one slot); restore_snapshot() brings it back with a device-to-device copy."""
        check(self._lib.gbp_ba_snapshot_state(self._h))

    def restore_snapshot(self):
        check(self._lib.gbp_ba_restore_snapshot(self._h))

    def save(self, path):
        np.save(path, self.save_state(), allow_pickle=False)

    def load(self, path):
        self.load_state(np.load(path, allow_pickle=False))

    def set_kernel_timing(self, on):
        """on = True / 1: events around every launch of the dominant kernel; on = n > 1: around every n-th launch."""
        check(self._lib.gbp_ba_set_kernel_timing(self._h, int(on)))

    def kernel_timing(self):
        ms, n, name = ct.c_double(), ct.c_int32(), ct.c_char_p()
        check(self._lib.gbp_ba_get_kernel_timing(self._h, ct.byref(ms), ct.byref(n), ct.byref(name)))
        return ms.value, n.value, (name.value or b'').decode()

    def kernel_times(self):
        """Milliseconds of every bracketed launch since set_kernel_timing, in order (call before kernel_timing())."""
        n = ct.c_int32()
        check(self._lib.gbp_ba_get_kernel_times(self._h, None, 0, ct.byref(n)))
        out = np.empty(n.value)
        check(self._lib.gbp_ba_get_kernel_times(self._h, dptr(out), n.value, ct.byref(n)))
        return out

    def sweep_clocks(self):
        """Device-clock stamps of every sweep since set_kernel_timing, microseconds since the first stamp: (n, 6) =
        fused sweep start, end | camera reduce start, end | camera finish start, end (NaN where a kernel did not run)."""
        n = ct.c_int32()
        check(self._lib.gbp_ba_get_sweep_clocks(self._h, None, 0, ct.byref(n)))
        out = np.full((n.value, 6), np.nan)
        if n.value:
            check(self._lib.gbp_ba_get_sweep_clocks(self._h, dptr(out), n.value, ct.byref(n)))
        return out

    def comm_info(self):
        k, r, n = ct.c_int32(), ct.c_int32(), ct.c_int32()
        check(self._lib.gbp_ba_comm_info(self._h, ct.byref(k), ct.byref(r), ct.byref(n)))
        return dict(kind=['none', 'callback', 'rccl', 'peer'][k.value], rank=r.value, n_ranks=n.value)

    def check_layout(self):
        """Debug kernel: number of slots that do not decode to the reference factor they hold (0 for a sound layout)."""
        v = ct.c_int32()
        check(self._lib.gbp_ba_check_layout(self._h, ct.byref(v)))
        return v.value

    def info(self):
        a, b, c = ct.c_int32(), ct.c_int32(), ct.c_int32()
        check(self._lib.gbp_ba_info(self._h, ct.byref(a), ct.byref(b), ct.byref(c)))
        return dict(fused=bool(a.value), cam_groups=a.value, n_tiles=b.value, n_blocks=c.value)

    def plan_info(self):
        """What the sweep's plan decided (gbp_ba_plan_info): which sweep and why, the SINGLE accumulation variant and its probe, the tile
        packing, camera windows (max_window = the largest per-workgroup camera set, 0: whole tables; table_rows; reduce_by_wave)."""
        v = np.zeros(_capi.PLAN_INFO_FIELDS, np.int32)
        check(self._lib.gbp_ba_plan_info(self._h, iptr(v), v.size))
        return dict(fused=bool(v[0]), staged_by_sparseness=bool(v[1]), single=bool(v[2]), single_probe=int(v[3]), pinned_tiles=int(v[4]),
                    n_blocks=int(v[5]), n_tiles=int(v[6]), pack_mode=int(v[7]), max_window=int(v[8]), table_rows=int(v[9]), reduce_by_wave=bool(v[10]))
