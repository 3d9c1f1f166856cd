from . import gbp
from . import gbp_ba
