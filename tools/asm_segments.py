#!/usr/bin/env python3
"""Per basic block of ONE kernel in a hipcc -S listing: MFMA / scratch load / scratch store / LDS / global counts, so that a
spill can be placed (hot loop or once-per-pass epilogue).  usage: asm_segments.py listing.s kernel_name_substring"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l)][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm')][0]
cnt = dict(mfma=0, sl=0, ss=0, ds=0, vm=0, valu=0)
def flush(i, l):
    print(f"{i-start:6d} {l[:44]:44s} " + " ".join(f"{k}={v}" for k, v in cnt.items()))
    for k in cnt: cnt[k] = 0
for i in range(start, end + 1):
    l = lines[i].strip()
    if re.match(r'^\.LBB\d+_\d+:', l) or 's_cbranch' in l or l.startswith('s_branch'):
        flush(i, l)
    if 'v_mfma' in l: cnt['mfma'] += 1
    elif l.startswith('scratch_load'): cnt['sl'] += 1
    elif l.startswith('scratch_store'): cnt['ss'] += 1
    elif l.startswith('ds_'): cnt['ds'] += 1
    elif l.startswith(('global_', 'buffer_')): cnt['vm'] += 1
    elif l.startswith('v_'): cnt['valu'] += 1
flush(end, 'end')
