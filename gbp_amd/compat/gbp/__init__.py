"""Drop-in `gbp` package: `gbp.gbp` (generic host graph) and `gbp.gbp_ba` (bundle adjustment on the MI355X engine)."""
from . import gbp, gbp_ba

__all__ = ['gbp', 'gbp_ba']
