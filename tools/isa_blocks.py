#!/usr/bin/env python3
"""Static instruction counts per basic block of one kernel in a hipcc -S listing.
usage: isa_blocks.py file.s kernel_substring [min_instrs]"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 25
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*' + re.escape(name) + r'\S*:', l))
blocks = []; cur = ['entry', 0, 0, 0, 0, 0]
for l in s[start + 1:]:
    t = l.strip()
    if t.startswith('.Lfunc_end'): break
    if not t or t.startswith(';'): continue
    m = re.match(r'^(\.LBB\S+):', t)
    if m:
        blocks.append(cur); cur = [m.group(1), 0, 0, 0, 0, 0]; continue
    if t.startswith('.'): continue
    op = t.split()[0]
    if op.startswith('v_'):
        cur[1] += 1
        if '_f64' in op: cur[5] += 1
    elif op.startswith('s_'): cur[2] += 1
    elif op.startswith('ds_'): cur[3] += 1
    else: cur[4] += 1
blocks.append(cur)
print('total valu %d salu %d ds %d mem %d f64 %d; blocks %d' % (*[sum(b[i] for b in blocks) for i in range(1, 6)], len(blocks)))
print('%-14s %5s %5s %4s %4s %5s' % ('block', 'valu', 'salu', 'ds', 'mem', 'f64'))
for b in blocks:
    if b[1] + b[2] >= thr: print('%-14s %5d %5d %4d %4d %5d' % tuple(b))
