#!/usr/bin/env python3
"""Instruction mix per straight-line segment (label / barrier to label / barrier) of one kernel in hipcc -S output.
usage: isa_mix.py file.s MANGLED_PREFIX [--dump]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and l.split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body = lines[start:end + 1]
def cls(op):
    if op.startswith('v_mfma'): return 'MFMA'
    if op.startswith('v_'): return 'VALU'
    if op.startswith('s_waitcnt'): return 'WAIT'
    if op.startswith('s_barrier'): return 'BAR'
    if op.startswith('s_'): return 'SALU'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'VMEM'
    return 'OTHER'
segs, cur, name = [], collections.Counter(), 'entry'
for l in body:
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') and not t.endswith(':'): continue
    if t.endswith(':'):
        segs.append((name, cur)); cur = collections.Counter(); name = t; continue
    c = cls(t.split()[0]); cur[c] += 1
    if c == 'BAR':
        segs.append((name, cur)); cur = collections.Counter(); name = name + '+bar'
segs.append((name, cur))
for n, c in segs:
    if sum(c.values()) > 3: print(f"{n[:44]:44s}", dict(c))
if '--dump' in sys.argv:
    print('\n'.join(body))
