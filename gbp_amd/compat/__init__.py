"""Drop-in packages with the reference's import names (`gbp`, `utils`, `vis`).

    import gbp_amd.compat; gbp_amd.compat.activate()      # or: PYTHONPATH=<repo>/gbp_amd/compat
    python /path/to/joeaortiz-gbp/ba.py --bal_file data/fr1desk_small.txt     # runs unchanged on the MI355X

`gbp.gbp_ba` (bundle adjustment: create_ba_graph / BAFactorGraph) is backed by the HIP engine through the C ABI.
`gbp.gbp` (FactorGraph / VariableNode / Factor with arbitrary Python measurement functions, used by
ndim_posegraph.py -- BASELINE.json config 1, "plumbing, no GPU") is host-side numpy.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def activate():
    """Put the drop-in packages (and the repo root, for gbp_amd itself) at the front of sys.path."""
    root = os.path.dirname(os.path.dirname(HERE))
    for p in (root, HERE):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
