#!/usr/bin/env python3
"""Wall time of the set-up path at the headline size (gbp_ba_create + priors + first beliefs), on the GPU box: from host arrays
(the ABI's default: includes the PCIe upload of the observations) and from arrays already on the device (GBP_FLAG_DEVICE_INPUT)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from gbp_amd.synthetic import make_synthetic
from gbp_amd.engine import BAEngine
t0 = time.perf_counter(); p = make_synthetic(n_cams=500, n_lmks=100_000, obs_per_lmk=10, seed=0); t1 = time.perf_counter()
print(f"make_synthetic {t1 - t0:.2f} s")
dev = torch.device('cuda', 0)
t = [torch.as_tensor(np.ascontiguousarray(x), device=dev) for x in (p.cam_means, p.lmk_means, p.meas, p.cam_idx.astype(np.int32), p.lmk_idx.astype(np.int32))]
torch.cuda.synchronize()
for name, mk in (('host arrays  ', lambda: BAEngine.from_problem(p)),
                 ('device arrays', lambda: BAEngine(p.K, *[x.data_ptr() for x in t], device_pointers=(p.n_cams, p.n_lmks, p.n_factors)))):
    for rep in range(4):
        t1 = time.perf_counter(); e = mk(); e.sync(); t2 = time.perf_counter()
        e.generate_priors_var(50.0); e.sync(); t3 = time.perf_counter()
        e.update_all_beliefs(); e.sync(); t4 = time.perf_counter()
        print(f"{name}: create {1e3 * (t2 - t1):7.2f} ms  generate_priors {1e3 * (t3 - t2):5.2f} ms  update_beliefs {1e3 * (t4 - t3):5.2f} ms", flush=True)
        e.close()
