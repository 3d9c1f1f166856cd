"""Per-basic-block instruction statistics of one kernel in a hipcc --save-temps .s file (where do the spills execute?).

usage: isa_blocks.py file.s kernel-name-substring"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(key) + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i] or lines[i].startswith('.Lfunc_end'))
blk, stats, order = 'entry', {}, []
for ln in lines[start + 1:end]:
    t = ln.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blk = m.group(1)
    if blk not in stats:
        stats[blk] = dict(n=0, spill_st=0, spill_ld=0, valu=0, vmem=0, lds=0, trans=0)
        order.append(blk)
    if not t or t[0] in '.;/':
        continue
    st = stats[blk]
    st['n'] += 1
    op = t.split()[0]
    st['spill_st'] += op.startswith('scratch_store')
    st['spill_ld'] += op.startswith('scratch_load')
    st['valu'] += op.startswith('v_')
    st['vmem'] += op.startswith(('global_', 'buffer_', 'flat_'))
    st['lds'] += op.startswith('ds_')
    st['trans'] += op.startswith(('v_rcp', 'v_sqrt', 'v_rsq', 'v_sin', 'v_cos'))
for b in order:
    st = stats[b]
    if st['n'] > 30 or st['spill_st'] or st['spill_ld']:
        print(f"{b:12s} " + ' '.join(f"{k}={v}" for k, v in st.items()))
tot = {k: sum(v[k] for v in stats.values()) for k in next(iter(stats.values()))}
print('total       ', ' '.join(f"{k}={v}" for k, v in tot.items()))
