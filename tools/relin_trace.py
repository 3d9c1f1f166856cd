import sys; sys.path.insert(0,'.')
from gbp_amd.synthetic import make_synthetic
from gbp_amd.engine import BAEngine
p = make_synthetic(n_cams=500, n_lmks=100_000, obs_per_lmk=10, seed=0)
e = BAEngine.from_problem(p); e.generate_priors_var(50.0); e.update_all_beliefs(); e.iterate(40)
print("relin no-reset:", list(e.relin_counts(40)))
