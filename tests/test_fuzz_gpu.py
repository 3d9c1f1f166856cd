"""Randomised shapes, GPU engine (both sweeps) against the CPU oracle: tiny and lopsided graphs, landmarks from degree 1
to several tiles, duplicate (camera, landmark) observations, every loss, random sweep flags.  Seeds are fixed: 0 .. 23, or
0 .. GBP_FUZZ_SEEDS - 1 for a longer soak (1000 seeds take a few minutes on the GPU box)."""
import os

import numpy as np
import pytest

from conftest import rel_err_rows
from gbp_amd.synthetic import BAProblem, make_synthetic

pytestmark = pytest.mark.gpu

COMPARED = {}
ESCAPES = []
N_SEEDS = int(os.environ.get('GBP_FUZZ_SEEDS', 24))


def random_problem(seed, min_deg=1):
    rng = np.random.default_rng(1000 + seed + (7919 if min_deg > 1 else 0))
    n_cams = int(rng.integers(2, 41))
    n_lmks = int(rng.integers(1, 240))
    base = make_synthetic(n_cams=max(n_cams, 12), n_lmks=n_lmks, obs_per_lmk=min(10, max(n_cams, 12)), seed=seed)
    cams = base.cam_means[:n_cams]
    ci, li, zz = [], [], []
    K = base.K
    from gbp_amd.synthetic import rodrigues
    for l in range(n_lmks):
        big = rng.random() < 0.04
        deg = int(rng.integers(65, 150)) if big else int(rng.integers(min_deg, max(min_deg, min(n_cams, 12)) + 1))
        if min_deg > 1 and rng.random() < 0.3:          # dense-packing variant: landmarks of 13 .. 63 factors, the shapes whole tiles fill badly
            deg = int(rng.integers(13, 64))
        cs = rng.integers(0, n_cams, size=deg) if (big or deg > n_cams or rng.random() < 0.2) else rng.choice(n_cams, size=deg, replace=False)
        for c in cs:                                   # (duplicates allowed: two factors between the same pair)
            R = rodrigues(cams[c, 3:])[0]
            y = R @ base.lmk_means[l] + cams[c, :3]
            if y[2] < 0.3:
                continue
            u = K[0] * y[0] / y[2] + K[2] + rng.normal(0, 1.5)
            v = K[1] * y[1] / y[2] + K[3] + rng.normal(0, 1.5)
            ci.append(c); li.append(l); zz.append((u, v))
    if not ci:
        ci, li, zz = [0], [0], [(320.0, 240.0)]
    order = rng.permutation(len(ci))
    used = np.unique(np.array(li))
    remap = -np.ones(n_lmks, dtype=np.int64); remap[used] = np.arange(used.size)
    ucam = np.unique(np.array(ci))
    cmap = -np.ones(n_cams, dtype=np.int64); cmap[ucam] = np.arange(ucam.size)
    return BAProblem(K=K, cam_means=cams[ucam], lmk_means=base.lmk_means[used], meas=np.array(zz)[order],
                     cam_idx=cmap[np.array(ci)[order]].astype(np.int32), lmk_idx=remap[np.array(li)[order]].astype(np.int32))


def healthy(o, p, are0):
    """The comparison is meaningful only while the run is sane.  GBP with aggressive settings can blow up (indefinite
    beliefs, ARE growing by orders of magnitude), and a cavity (belief minus own message) can come arbitrarily close to
    singular -- from there two correct implementations (LDL^T here, Gauss-Jordan in the oracle, LU in numpy) differ by
    cond x eps, amplified by every following sweep."""
    from gbp_amd.balio import reference_factor_order
    order = reference_factor_order(p.cam_idx)
    cam, lmk = p.cam_idx[order], p.lmk_idx[order]
    _, cl, _, ll = o.beliefs()
    _, mcl, _, mll = o.messages()
    ec = np.linalg.eigvalsh(cl[cam] - mcl)
    el = np.linalg.eigvalsh(ll[lmk] - mll)
    ok = ec.min() > 0 and el.min() > 0 and (ec[:, -1] / ec[:, 0]).max() < 1e7 and (el[:, -1] / el[:, 0]).max() < 1e7
    # a landmark drifting into a camera's focal plane (depth -> 0) makes the projection and its Jacobian ill-conditioned:
    # the factor becomes ~1e7 times stronger than the cavities and the Schur complement cancels that many digits (seed 11:
    # depth 0.037, 1e-12 differences in the means became 1e-4 in one message)
    cm, lm = o.means()
    from gbp_amd.synthetic import rodrigues
    pc = np.einsum('fij,fj->fi', rodrigues(cm[cam, 3:]), lm[lmk]) + cm[cam, :3]
    ok = ok and (np.abs(pc[:, 2]) / np.linalg.norm(pc, axis=1)).min() > 0.2
    return bool(ok) and o.are() < 1e3 * max(are0, 1.0)


def third_opinion(p, cfg, flags_done, o):
    """(belief spread, energy spread) between the C oracle and a third implementation of the same sweeps -- the object-per-
    factor numpy restatement, np.linalg.inv like the reference.  The health filter above cannot see every ill-conditioned
    state (2 of 1500 seeds: a relinearisation with residuals of ~40 px moves all three implementations 1e-6 apart from each
    other in one sweep, tools/fuzz_diag.py): a GPU-oracle gap is a failure only if two CPU implementations agree better."""
    from oracle.numpy_ba import NumpyBA
    nb = NumpyBA(p, **cfg)
    nb.generate_priors_var(30.0)
    nb.update_all_beliefs()
    for rob, rel in flags_done:
        nb.synchronous_iteration(robustify=rob, local_relin=rel)
    spread = max(rel_err_rows(a, b) for a, b in zip(nb.beliefs(), o.beliefs()))
    return spread, abs(nb.energy() - o.energy()) / max(abs(o.energy()), 1e-300)


@pytest.mark.parametrize('seed', range(N_SEEDS))
def test_random_shapes_against_oracle(oracle_mod, seed):
    run_seed(oracle_mod, seed, random_problem(seed), seed)


@pytest.mark.parametrize('seed', range(max(1, N_SEEDS // 2)))
def test_random_shapes_dense_packing(oracle_mod, monkeypatch, seed):
    """The same comparison on graphs whose landmarks have three or more factors (a third of them 13 .. 63, some above 64), with the
    dense tile packing asked for (GBP_PACK=dense: tile t = factors [64 t, 64 t + 64), landmarks span tiles, partial sums +
    k_lmk_finish_parts).  A graph in which an observation fell behind its camera and left a landmark with two factors gets the
    whole-landmark packing instead -- also fine."""
    monkeypatch.setenv('GBP_PACK', 'dense')
    run_seed(oracle_mod, 100_000 + seed, random_problem(seed, min_deg=3), seed)


@pytest.mark.parametrize('seed', range(max(1, N_SEEDS // 2)))
def test_random_shapes_camera_windows(oracle_mod, monkeypatch, seed):
    """The same comparison with per-workgroup camera windows asked for wherever they fit (GBP_WINDOWS=1): every workgroup's table covers
    the interval of cameras its own tiles meet, whatever its width -- here one or a few tiles per workgroup on 2 .. 40 random cameras --
    and the reduce adds a camera's rows through the per-camera row ranges (odd seeds: its tree form, and the dense packing)."""
    monkeypatch.setenv('GBP_WINDOWS', '1')
    if seed % 2:
        monkeypatch.setenv('GBP_ROWS_WAVE_MAX', '0')
        monkeypatch.setenv('GBP_PACK', 'dense')
    run_seed(oracle_mod, 200_000 + seed, random_problem(seed, min_deg=3 if seed % 2 else 1), seed, want_windows=True)


def run_seed(oracle_mod, key, p, seed, want_windows=False):
    from gbp_amd.engine import BAEngine
    rng = np.random.default_rng(seed)
    loss = [None, 'huber', 'constant'][seed % 3]
    # thresholds away from the degenerate beta = 0 (there "relinearise iff distance > 0" flips on the last bit of a solve)
    cfg = dict(loss=loss, Nstds=float(rng.uniform(1.0, 3.0)), beta=float(rng.choice([0.005, 0.01, 0.05])),
               num_undamped_iters=int(rng.choice([1, 2, 6])), min_linear_iters=int(rng.choice([2, 4, 8])),
               eta_damping=float(rng.choice([0.3, 0.4, 0.7])), gauss_noise_std=float(rng.uniform(1.5, 3.0)))
    flags = [(bool(rng.integers(0, 2)), bool(rng.random() < 0.8)) for _ in range(8)]
    o = oracle_mod.OracleBA.from_problem(p, threads=4, **cfg)
    engines = [BAEngine.from_problem(p, fused=True, **cfg), BAEngine.from_problem(p, fused=False, **cfg)]
    if want_windows:
        assert engines[0].plan_info()['max_window'] > 0, engines[0].plan_info()
    for g in [o] + engines:
        g.generate_priors_var(30.0)
        g.update_all_beliefs()
    are0 = o.are()
    compared = 0
    for k, (rob, rel) in enumerate(flags):
        for g in [o] + engines:
            g.synchronous_iteration(robustify=rob, local_relin=rel)
        if not healthy(o, p, are0):
            break
        ob, so = o.beliefs(), o.relin_state()
        gaps = [max(rel_err_rows(a, b) for a, b in zip(e.beliefs(), ob)) for e in engines]
        # (abs: graphs of a handful of factors can be fitted almost exactly -- residuals ~1e-3 px are differences of numbers
        #  ~300 that agree to 1e-10 between the two implementations)
        egaps = [abs(e.energy() - o.energy()) for e in engines]
        etol = max(1e-6 * abs(o.energy()), 1e-8)
        if max(gaps) >= 1e-6 or max(egaps) > etol:
            spread, espread = third_opinion(p, cfg, flags[:k + 1], o)
            assert max(gaps) < max(1e-6, 4.0 * spread), (seed, compared, gaps, spread, p.n_cams, p.n_lmks, p.n_factors)
            assert max(egaps) <= max(etol, 4.0 * espread * abs(o.energy())), (seed, 'energy', egaps, espread)
            ESCAPES.append(key)                            # (bounded in test_fuzz_was_not_vacuous: the escape must stay the exception)
            break                                           # ill-conditioned from here on: nothing more to learn from this seed
        for e in engines:
            se = e.relin_state()
            assert np.array_equal(so['iters_since_relin'], se['iters_since_relin'])
            assert np.array_equal(so['robust_flag'], se['robust_flag'])
            assert np.allclose(so['adaptive_var'], se['adaptive_var'], rtol=1e-6)
        compared += 1
    assert compared >= 1, (key, compared)
    COMPARED[key] = compared


def test_fuzz_was_not_vacuous():
    """Most sweeps of most seeds must have been comparable (the health filter may only cut the odd blown-up run short)."""
    if os.environ.get('PYTEST_XDIST_WORKER'):
        pytest.skip('the seeds are spread over xdist workers (soak runs): the tally lives in the other processes')
    n = N_SEEDS + 2 * max(1, N_SEEDS // 2)               # plain + dense packing + camera windows
    assert len(COMPARED) == n and sum(COMPARED.values()) >= 0.6 * 8 * n, COMPARED
    # the third-opinion escape (a GPU-oracle gap above 1e-6 excused by an equally large gap between two CPU implementations) is for the
    # odd ill-conditioned relinearisation: at most one seed in 24 may take it, everything before that sweep was held to 1e-6
    assert len(ESCAPES) <= max(1, n // 24), ESCAPES
