#!/usr/bin/env python3
"""profiles/<tag>_kernel_resource_usage.txt: -Rpass-analysis=kernel-resource-usage of the sweep / reduce / general kernels of the
library as it is built (registers, spills, occupancy, LDS) -- one line per kernel, demangled.  No GPU needed.

    python tools/resource_usage.py r04
"""
import os, re, subprocess, sys
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, REPO)
from gbp_amd import build
tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
_, err = build.build(force=True, out='/tmp/_usage.so', extra_flags=['-Rpass-analysis=kernel-resource-usage'], capture=True)
rows = {}
for blk in re.split(r'remark: Function Name: ', err)[1:]:
    name = blk.split()[0]
    get = lambda k: re.search(re.escape(k) + r': (\d+)', blk)
    keys = ['TotalSGPRs', 'VGPRs', 'AGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'SGPRs Spill', 'VGPRs Spill', 'LDS Size [bytes/block]']
    if not all(get(k) for k in keys):
        continue
    dem = subprocess.run(['c++filt', name], stdout=subprocess.PIPE, text=True).stdout.strip().split('(')[0]
    if any(s in dem for s in ('k_sweep_wat', 'k_cam_reduce', 'k_cam_partial_staged', 'k_factor_tile')):
        rows[dem] = f"{dem}: " + ', '.join(f"{k}={get(k).group(1)}" for k in keys)
out = os.path.join(REPO, 'profiles', f'{tag}_kernel_resource_usage.txt')
open(out, 'w').write('\n'.join(rows[k] for k in sorted(rows)) + '\n')
print(open(out).read())
