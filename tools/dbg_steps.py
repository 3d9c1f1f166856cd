import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from gbp_amd.balio import read_bal
from gbp_amd.engine import BAEngine
p = read_bal('tests/golden/data/fr1desk_vsmall.txt')
e = BAEngine.from_problem(p, fused=False); e.sync(); print('create ok', e.info(), flush=True)
print(e.residual_sums(), flush=True)
