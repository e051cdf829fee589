"""per-basic-block instruction counts of a kernel listing dumped by tools/asm_stats.py --dump (find the hot loop)"""
import re
import sys
from collections import Counter

QUARTER = ('v_mul_lo_u32', 'v_mul_hi_u32', 'v_sqrt_f32', 'v_rcp_f32', 'v_mad_u64_u32', 'v_mad_i64_i32', 'v_rsq_f32')
lines = open(sys.argv[1]).read().split('\n')
blocks = []
cur = ['entry', []]
for l in lines:
    t = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append(cur)
        cur = [m.group(1), []]
        continue
    if t.startswith('; %bb.'):
        blocks.append(cur)
        cur = [t.split(':')[0][2:], []]
        continue
    if t and not t.startswith((';', '.')) and not t.endswith(':'):
        cur[1].append(t)
blocks.append(cur)
for name, ins in blocks:
    c = Counter(i.split()[0] for i in ins)
    v = sum(n for k, n in c.items() if k.startswith('v_'))
    sa = sum(n for k, n in c.items() if k.startswith('s_') and k not in ('s_waitcnt', 's_nop'))
    m = sum(n for k, n in c.items() if 'load' in k or 'store' in k)
    br = [i for i in ins if i.startswith(('s_cbranch', 's_branch'))]
    print('%-12s n=%4d valu=%4d pk=%3d mov=%3d cnd=%3d salu=%3d vmem=%2d f64=%2d qr=%2d  %s' % (
        name, len(ins), v, sum(n for k, n in c.items() if k.startswith('v_pk')),
        sum(n for k, n in c.items() if k.startswith('v_mov')), sum(n for k, n in c.items() if 'cndmask' in k), sa, m,
        sum(n for k, n in c.items() if 'f64' in k), sum(n for k, n in c.items() if k.split('_e')[0] in QUARTER),
        ' '.join(br)))
