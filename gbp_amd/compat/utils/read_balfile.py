"""BAL-style reader with the reference's return signature (utils/read_balfile.py:4-37)."""
import numpy as np

from gbp_amd.balio import read_bal


def read_balfile(balfile):
    p = read_bal(balfile)
    K = np.array([[p.K[0], 0.0, p.K[2]], [0.0, p.K[1], p.K[3]], [0.0, 0.0, 1.0]])
    return (p.n_cams, p.n_lmks, p.n_factors, p.cam_means, p.lmk_means, p.meas,
            [int(c) for c in p.cam_idx], [int(l) for l in p.lmk_idx], K)
