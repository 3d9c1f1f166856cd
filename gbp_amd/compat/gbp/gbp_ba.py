"""Bundle adjustment with GBP on the MI355X behind the reference's `gbp.gbp_ba` names.

`create_ba_graph(bal_file, configs)` returns a BAFactorGraph whose sweep, belief updates, priors and diagnostics are
HIP kernels (gbp_amd.engine.BAEngine -> include/gbp_ba.h).  `graph.cam_nodes`, `graph.lmk_nodes` and `graph.factors`
are lazy sequences of thin views over host mirrors of the device state, so scripts written for joeaortiz/gbp
(ba.py:68-105, vis/ba_vis.py:35-55,105-115) run unchanged -- including per-factor writes to `iters_since_relin`.
There is no CPU implementation behind this module: without a gfx950 device create_ba_graph raises.
"""
from array import array

import numpy as np

from gbp_amd.balio import read_bal, reference_factor_order
from gbp_amd.engine import BAEngine
from utils.gaussian import NdimGaussian


class _Lazy:
    """Sequence of views created on demand (1M Python objects up front would cost more than the solve)."""

    def __init__(self, n, make):
        self._n, self._make, self._cache = n, make, {}

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        v = self._cache.get(i)
        if v is None:
            v = self._cache[i] = self._make(i)
        return v

    def __iter__(self):
        return (self[i] for i in range(self._n))

    def __add__(self, other):
        return list(self) + list(other)


class _Concat:
    """cam_nodes followed by lmk_nodes without materialising either (graph.var_nodes, gbp_ba.py:147)."""

    def __init__(self, a, b):
        self._a, self._b = a, b

    def __len__(self):
        return len(self._a) + len(self._b)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return self._a[i] if i < len(self._a) else self._b[i - len(self._a)]

    def __iter__(self):
        yield from self._a
        yield from self._b


class _PriorProxy(NdimGaussian):
    """node.prior: reads come from the device state; assigning .eta / .lam (what scripts written for the reference do,
    e.g. ndim_posegraph.py:51-52) is kept on the host and uploaded before the next device call."""

    def __init__(self, graph, kind, index, eta, lam):
        object.__setattr__(self, '_bind', None)
        super().__init__(len(eta), eta, lam)
        object.__setattr__(self, '_bind', (graph, kind, index))

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        bind = self.__dict__.get('_bind')
        if bind is not None and name in ('eta', 'lam'):
            bind[0]._write_prior(bind[1], bind[2], name, value)


class _VariableView:
    """VariableNode surface (gbp.py:156-198) of one camera or landmark."""

    def __init__(self, graph, kind, index):
        self._g, self._kind, self._i = graph, kind, index
        self.variableID = index if kind == 0 else graph._C + index
        self.dofs = 6 if kind == 0 else 3
        self.prior_lambda_end = -1
        self.prior_lambda_logdiff = -1
        if kind == 0:
            self.c_id = index
        else:
            self.l_id = index

    @property
    def mu(self):
        return self._g._means()[self._kind][self._i]

    @property
    def Sigma(self):
        return self._g._covs()[self._kind][self._i]

    @property
    def belief(self):
        b = self._g._beliefs()
        return NdimGaussian(self.dofs, b[2 * self._kind][self._i], b[2 * self._kind + 1][self._i])

    @property
    def prior(self):
        b = self._g._priors()
        return _PriorProxy(self._g, self._kind, self._i, b[2 * self._kind][self._i], b[2 * self._kind + 1][self._i])

    @property
    def adj_factors(self):
        return [self._g.factors[int(f)] for f in self._g._adjacent(self._kind, self._i)]


class _FactorViewBase(int):
    """Factor surface (gbp.py:201-249) of one reprojection factor (reference factor order).  A view IS its factor id (an int with no
    state of its own: everything lives in the graph), and `iters_since_relin` -- the one attribute ba.py touches on EVERY factor in
    EVERY iteration (ba.py:91-93 writes, :96-99 reads) -- is a property whose getter and setter are the C-level `__getitem__` /
    `__setitem__` of the graph's host mirror (an int32 array.array): no Python frame per factor.  The class is specialised per graph (_factor_view_class)."""
    __slots__ = ()
    _g = None
    dofs_conditional_vars = 9

    @property
    def factorID(self):
        return int(self)

    @property
    def args(self):
        return (self._g._K,)                             # vis/ba_vis.py:115 reads K from factors[0].args[0]

    @property
    def loss(self):
        return self._g._configs.get('loss')

    @property
    def mahalanobis_threshold(self):
        return self._g._configs.get('Nstds')

    @property
    def gauss_noise_var(self):
        return float(self._g._configs['gauss_noise_std']) ** 2

    @property
    def eta_damping(self):
        return float(self._g._relin()['eta_damping'][self])

    @property
    def adaptive_gauss_noise_var(self):
        return float(self._g._relin()['adaptive_var'][self])

    @property
    def robust_flag(self):
        return bool(self._g._relin()['robust_flag'][self])

    @property
    def adj_vIDs(self):
        c, l = self._g._cam_of[self], self._g._lmk_of[self]
        return [int(c), int(self._g._C + l)]

    @property
    def adj_var_nodes(self):
        return [self._g.cam_nodes[int(self._g._cam_of[self])], self._g.lmk_nodes[int(self._g._lmk_of[self])]]

    @property
    def adj_beliefs(self):
        return [n.belief for n in self.adj_var_nodes]

    @property
    def measurement(self):
        return self._g._lin(int(self))['z']

    @property
    def linpoint(self):
        return self._g._lin(int(self))['linpoint']

    @property
    def factor(self):
        d = self._g._engine.factors(int(self), 1)
        return NdimGaussian(9, d['eta'][0], d['lam'][0])

    @property
    def messages(self):
        ce, cl, le, ll = self._g._engine.messages(int(self), 1)
        return [NdimGaussian(6, ce[0], cl[0]), NdimGaussian(3, le[0], ll[0])]

    def compute_residual(self):
        from gbp.factors import reprojection
        x = np.concatenate([n.mu for n in self.adj_var_nodes])
        return reprojection.meas_fn(x, self._g._K) - self.measurement

    def reprojection_err(self):
        return float(np.linalg.norm(self.compute_residual()))


def _factor_view_class(graph, mirror):
    return type('_FactorView', (_FactorViewBase,), dict(__slots__=(), _g=graph, iters_since_relin=property(mirror.__getitem__, mirror.__setitem__)))


class _FactorSeq:
    """graph.factors: F stateless views over ONE host mirror of iters_since_relin (an int32 array.array: its C-level __getitem__ /
    __setitem__ are the views' property, and numpy reads or overwrites all of it through the buffer protocol in about a millisecond per
    million factors).  The mirror is kept current for as long as views may be around: the graph re-reads it after every device call
    that can change it (BAFactorGraph._invalidate), so a view HELD across synchronous_iteration() reads the new value like the
    reference's long-lived Factor objects do, and whatever was written through any view is found by comparing the mirror with the
    device's values before the next device call (BAFactorGraph._flush)."""

    def __init__(self, graph, n):
        self._g, self._n = graph, n
        self._mirror = array('i', bytes(4 * n))
        self._np = np.frombuffer(self._mirror, dtype=np.int32) if n else np.zeros(0, np.int32)      # the same memory, as numpy sees it
        self._cls = _factor_view_class(graph, self._mirror)
        self._views = None

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        self._g._refresh_iters()
        return self._cls(i)

    def __iter__(self):
        self._g._refresh_iters()
        if self._views is None:
            self._views = list(map(self._cls, range(self._n)))
        return iter(self._views)

    def __add__(self, other):
        return list(self) + list(other)


class BAFactorGraph:
    """gbp/gbp_ba.py:12-69 + the FactorGraph methods ba.py uses (gbp.py:36-92), on the GPU."""

    def __init__(self, problem, configs, device=0):
        self._configs = dict(configs)
        self._engine = BAEngine.from_problem(
            problem, gauss_noise_std=float(configs['gauss_noise_std']), loss=configs.get('loss'),
            Nstds=float(configs.get('Nstds', 3.0)), beta=float(configs['beta']),
            num_undamped_iters=int(configs['num_undamped_iters']), min_linear_iters=int(configs['min_linear_iters']),
            eta_damping=float(configs['eta_damping']), device=device)
        self._C, self._L, self._F = problem.n_cams, problem.n_lmks, problem.n_factors
        self._K = np.array([[problem.K[0], 0.0, problem.K[2]], [0.0, problem.K[1], problem.K[3]], [0.0, 0.0, 1.0]])
        order = reference_factor_order(problem.cam_idx)
        self._cam_of, self._lmk_of = problem.cam_idx[order], problem.lmk_idx[order]
        self.nonlinear_factors = True
        self.eta_damping = configs['eta_damping']
        self.beta = configs['beta']
        self.num_undamped_iters = configs['num_undamped_iters']
        self.min_linear_iters = configs['min_linear_iters']
        self.cam_nodes = _Lazy(self._C, lambda i: _VariableView(self, 0, i))
        self.lmk_nodes = _Lazy(self._L, lambda i: _VariableView(self, 1, i))
        self.factors = _FactorSeq(self, self._F)
        self.var_nodes = _Concat(self.cam_nodes, self.lmk_nodes)
        self.n_var_nodes, self.n_factor_nodes, self.n_edges = self._C + self._L, self._F, 2 * self._F
        self._cache = {}
        self._iters_dev, self._iters_fresh = None, False  # iters_since_relin as last read from the device; is the factors' mirror current?
        self._views_out = False                           # has a factor view ever been handed out?  (then the mirror is kept current)
        self._priors_host = None                          # priors written from Python, waiting to go to the device
        self._adj = None

    # ---- host mirrors ------------------------------------------------------------------------------------------
    def _invalidate(self, iters_changed=True):
        """After a device call.  Once factor views are out the mirror of iters_since_relin is re-read at once (4 bytes per factor): a view
        the caller still holds must show the device's value, and a write through it must not be lost to a later refresh."""
        self._cache.clear()
        if iters_changed:
            self._iters_fresh = False
            if self._views_out:
                self._refresh_iters()

    def _refresh_iters(self):
        """graph.factors' mirror of iters_since_relin: one 4 F-byte read per device state, shared by every view"""
        self._views_out = True
        if not self._iters_fresh:
            self._iters_dev = self._engine.iters_since_relin()
            self.factors._np[:] = self._iters_dev
            self._iters_fresh = True

    def _cached(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    def _means(self):
        return self._cached('mu', self._engine.means)

    def _beliefs(self):
        return self._cached('bel', self._engine.beliefs)

    def _priors(self):
        if self._priors_host is not None:
            return self._priors_host
        return self._cached('pri', self._engine.priors)

    def _write_prior(self, kind, i, name, value):
        if self._priors_host is None:
            self._priors_host = [np.array(a) for a in self._engine.priors()]
        self._priors_host[2 * kind + (name == 'lam')][i] = np.asarray(value, dtype=np.float64)

    _LIN_MIRROR_MAX = 200_000      # factors: up to here measurement / linpoint reads share one host mirror of all factors

    def _lin(self, f):
        """Factor.measurement / Factor.linpoint: a mirror of all factors for graphs a Python loop can walk, a one-factor
        device gather (gbp_ba_get_factors moves only the requested range) above that."""
        if self._F <= self._LIN_MIRROR_MAX:
            d = self._cached('lin', lambda: self._engine.factors(dense=False))
            return dict(z=d['z'][f], linpoint=d['linpoint'][f])
        d = self._engine.factors(f, 1, dense=False)
        return dict(z=d['z'][0], linpoint=d['linpoint'][0])

    def _covs(self):
        return self._cached('cov', self._engine.covariances)

    def _relin(self):
        self._flush()
        return self._cached('relin', self._engine.relin_state)

    def _flush(self):
        if self._priors_host is not None:
            self._engine.set_priors(*self._priors_host)
            self._priors_host = None
            self._cache.pop('pri', None)
        if self._iters_fresh and not np.array_equal(self.factors._np, self._iters_dev):      # written through a view (ba.py:91-93)
            it = self.factors._np
            if np.all(it == it[0]):
                self._engine.set_iters_since_relin(int(it[0]))          # ba.py:91-93 writes the same value everywhere
            else:
                self._engine.set_iters_since_relin(it)
            self._iters_dev = it.copy()
            self._cache.pop('relin', None)

    def _adjacent(self, kind, i):
        if self._adj is None:
            self._adj = (np.argsort(self._cam_of, kind='stable'), np.searchsorted(np.sort(self._cam_of), np.arange(self._C + 1)),
                         np.argsort(self._lmk_of, kind='stable'), np.searchsorted(np.sort(self._lmk_of), np.arange(self._L + 1)))
        order, ptr = (self._adj[0], self._adj[1]) if kind == 0 else (self._adj[2], self._adj[3])
        return order[ptr[i]:ptr[i + 1]]

    # ---- priors (gbp_ba.py:20-52) ------------------------------------------------------------------------------
    def generate_priors_var(self, weaker_factor=100):
        self._flush()
        self._engine.generate_priors_var(weaker_factor)
        self._invalidate(iters_changed=False)

    def weaken_priors(self, weakening_factor):
        self._flush()                     # priors written through node.prior first, or the next flush would undo the weakening
        self._engine.weaken_priors(weakening_factor)
        self._invalidate(iters_changed=False)

    def set_priors_var(self, priors):
        self._flush()
        self._engine.set_priors_var(priors)
        self._invalidate(iters_changed=False)

    # ---- sweep (gbp.py:46-92) ----------------------------------------------------------------------------------
    def update_all_beliefs(self):
        self._flush()
        self._engine.update_all_beliefs()
        self._invalidate(iters_changed=False)

    def synchronous_iteration(self, local_relin=True, robustify=False):
        self._flush()
        self._engine.synchronous_iteration(local_relin=local_relin, robustify=robustify)
        self._invalidate()

    def iterate(self, n, robustify=True, local_relin=True):
        """n sweeps in one call (no reference counterpart: saves the per-call host overhead)."""
        self._flush()
        self._engine.iterate(n, robustify=robustify, local_relin=local_relin)
        self._invalidate()

    # The reference lets a caller run the four stages of a sweep one by one (gbp.py:46-84); on the device each is a stage kernel on
    # the same state (include/gbp_ba.h).  A relinearisation decided by relinearise_factors / compute_all_factors shows in
    # factor.linpoint / factor.factor at once and enters the messages when they are next computed.
    def robustify_all_factors(self):
        self._flush()
        self._engine.robustify_all_factors()
        self._invalidate(iters_changed=False)

    def relinearise_factors(self):
        self._flush()
        self._engine.relinearise_factors()
        self._invalidate()

    def compute_all_messages(self, local_relin=True):
        self._flush()
        self._engine.compute_all_messages(local_relin)
        self._invalidate()

    def compute_all_factors(self):
        self._flush()
        self._engine.compute_all_factors()
        self._invalidate()

    def joint_distribution_inf(self):
        """gbp.py:94-134: joint (eta, Lambda) over all variables from the priors and the factors at their current
        linearisation points -- dense, for graphs small enough to solve in batch."""
        n = 6 * self._C + 3 * self._L
        if n > 20000:
            raise MemoryError(f"the dense joint of {n} scalar variables does not fit a batch solve")
        self._flush()
        eta, lam = np.zeros(n), np.zeros((n, n))
        ce, cl, le, ll = self._priors()
        for c in range(self._C):
            eta[6 * c:6 * c + 6] += ce[c]; lam[6 * c:6 * c + 6, 6 * c:6 * c + 6] += cl[c]
        o = 6 * self._C
        for l in range(self._L):
            eta[o + 3 * l:o + 3 * l + 3] += le[l]; lam[o + 3 * l:o + 3 * l + 3, o + 3 * l:o + 3 * l + 3] += ll[l]
        fac = self._engine.factors()
        for f in range(self._F):
            a, b = 6 * int(self._cam_of[f]), o + 3 * int(self._lmk_of[f])
            fe, fl = fac['eta'][f], fac['lam'][f]
            eta[a:a + 6] += fe[:6]; eta[b:b + 3] += fe[6:]
            lam[a:a + 6, a:a + 6] += fl[:6, :6]; lam[a:a + 6, b:b + 3] += fl[:6, 6:]
            lam[b:b + 3, a:a + 6] += fl[6:, :6]; lam[b:b + 3, b:b + 3] += fl[6:, 6:]
        return eta, lam

    def joint_distribution_cov(self):
        eta, lam = self.joint_distribution_inf()
        sigma = np.linalg.inv(lam)
        return sigma @ eta, sigma

    # ---- diagnostics (gbp_ba.py:54-69, gbp.py:36-44) -----------------------------------------------------------
    def are(self):
        self._flush()
        return self._engine.are()

    def energy(self):
        self._flush()
        return self._engine.energy()

    def count_relinearising(self):
        """ba.py:96-99 (`sum(factor.iters_since_relin == 0 for factor in graph.factors)`) as one device reduction."""
        self._flush()
        return self._engine.count_relinearising()

    def compute_residuals(self):
        out = []
        for f in self.factors:
            out += list(f.compute_residual())
        return out

    def get_means(self):
        cm, lm = self._means()
        return np.concatenate([cm.reshape(-1), lm.reshape(-1)])


# the reference's class names, kept importable
ReprojectionFactor = _FactorViewBase
FrameVariableNode = _VariableView
LandmarkVariableNode = _VariableView


def create_ba_graph(bal_file, configs, device=0):
    """gbp/gbp_ba.py:97-150: read the BAL-style file and build the graph (factors camera-major, linearised at the file means)."""
    return BAFactorGraph(read_bal(bal_file), configs, device=device)
