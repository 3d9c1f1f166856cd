import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gbp_amd.synthetic import make_synthetic
from gbp_amd.engine import BAEngine
for n_cams, n_lmks in ((60, 4000), (500, 20000)):
    p = make_synthetic(n_cams=n_cams, n_lmks=n_lmks, obs_per_lmk=10, seed=2)
    ref = BAEngine.from_problem(p); ref.generate_priors_var(50.0); ref.update_all_beliefs(); ref.iterate(12)
    rb = ref.beliefs()
    e = BAEngine.from_problem(p); e.peer_connect(0, [e.peer_export(1)])
    e.generate_priors_var(50.0); e.update_beliefs_sharded(); e.iterate_sharded(12); e.sync()
    eb = e.beliefs()
    print(n_cams, n_lmks, "single-process merged peer vs engine: bitwise", all(np.array_equal(a, b) for a, b in zip(rb, eb)),
          "max abs", max(np.abs(a - b).max() for a, b in zip(rb, eb)))
