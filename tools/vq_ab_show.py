#!/usr/bin/env python3
"""Compact view of a tools/ubench/vq_ab output file (best / median us per build)."""
import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{"rows"'):
        d = json.loads(line)
        print(d['rows'], d['form'], '  '.join(f"{k[9:-3]}:{v['best_us']:.1f}/{v['median_us']:.1f}{'' if v['bit_exact'] else ' BAD'}" for k, v in d.items() if k.startswith('lib')))
    elif len(sys.argv) > 2 or not line.startswith(' '):
        print(line.rstrip()[:330])
