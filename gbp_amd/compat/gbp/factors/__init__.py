from . import reprojection
from . import linear_displacement
