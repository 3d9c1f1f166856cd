"""Factor models with the reference's module names (`from gbp.factors import reprojection, linear_displacement`)."""
from . import linear_displacement, reprojection

__all__ = ['linear_displacement', 'reprojection']
