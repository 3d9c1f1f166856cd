"""The linear-GBP oracle (oracle/linear_oracle.py) against fixture G8 = the reference's own ndim_posegraph.py run
(--n_varnodes 100 --dim 3 --n_iters 20): energies, distances to the batch MAP and final means."""
import numpy as np

from conftest import golden
from oracle.linear_oracle import LinearOracle, toy_posegraph


def test_linear_oracle_reproduces_reference_trace():
    g8 = golden('G8_toy_linear')
    va, vb, fe, fl, fc, pe, pl = toy_posegraph(100, 3, 10, 1.0, seed=0)
    o = LinearOracle(va, vb, fe, fl, pe, pl, factor_const=fc)
    o.update_all_beliefs()
    mu_map = g8['n100d3_map_mu']
    energy, dist = [], []
    for _ in range(20):
        o.synchronous_iteration()
        energy.append(o.energy())
        dist.append(np.linalg.norm(o.get_means() - mu_map))
    assert np.allclose(energy, g8['n100d3_energy'], rtol=1e-6, atol=1e-3)      # fixture values are the printed 4 decimals
    assert np.allclose(dist, g8['n100d3_dist'], rtol=1e-5, atol=1e-5)
    assert np.allclose(o.get_means(), g8['n100d3_final_means'], rtol=1e-9, atol=1e-9)


def test_linear_oracle_damping_and_energy_identity():
    """Damped messages mix with the old eta only (gbp.py:368); the (eta_f, Lambda_f, const) energy equals the residual form."""
    va, vb, fe, fl, fc, pe, pl = toy_posegraph(12, 2, 3, 0.5, seed=1)
    o = LinearOracle(va, vb, fe, fl, pe, pl, factor_const=fc, eta_damping=0.4)
    o.update_all_beliefs()
    o.iterate(3)
    e = 0.0
    rs = np.random.RandomState(1)                 # regenerate the measurements the same way toy_posegraph did
    mu0 = rs.rand(12, 2) * 10
    z = []
    pairs = []
    for i, m in enumerate(mu0):
        d = np.array([np.linalg.norm(m - m1) for m1 in mu0])
        for j in d.argsort()[1:4]:
            if [j, i] not in pairs:
                z.append(m - mu0[j] + rs.normal(0., 0.5, 2)); pairs.append([i, j])
    for f, (i, j) in enumerate(pairs):
        r = (o.mu[j] - o.mu[i]) - z[f]
        e += 0.5 * r @ r / 0.25
    assert np.isclose(o.energy(), e, rtol=1e-10)
