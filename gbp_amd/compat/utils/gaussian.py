"""Information-form Gaussian container (the reference's utils/gaussian.py:4-16 interface)."""
import numpy as np


class NdimGaussian:
    """eta (dim,) and lam (dim, dim); anything missing or mis-shaped starts as zeros."""

    def __init__(self, dimensionality, eta=None, lam=None):
        n = int(dimensionality)
        self.dim = n
        ok_eta = eta is not None and len(eta) == n
        ok_lam = lam is not None and getattr(lam, 'shape', None) == (n, n)
        self.eta = eta if ok_eta else np.zeros(n)
        self.lam = lam if ok_lam else np.zeros((n, n))
