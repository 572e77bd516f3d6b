#!/usr/bin/env python3
"""Scans gfx950 assembly (hipcc -S) for a store-data hazard hipcc (ROCm 7.2) does not guard: a buffer store of more than
8 bytes whose soffset is an SGPR, followed within two instructions by a vector instruction that writes one of its data
registers.  LLVM's hazard recognizer exempts the SGPR-soffset form (GCNHazardRecognizer::createsVALUHazard); on MI355X the
overwrite corrupts the last dword of the last lanes of each row (found with vq_track_kernel_d64's z_q stores, round 3).
usage: hazard_scan.py file.s [...]   exit code 1 if a site is found."""
import re
import sys


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    lines = [l.strip() for l in open(path)]
    ins = [(i, l) for i, l in enumerate(lines) if l and not l.startswith((';', '.', '/')) and not l.endswith(':')]
    found = 0
    for n, (i, l) in enumerate(ins):
        m = re.match(r'buffer_store_dwordx[34]\s+(v\[\d+:\d+\]),\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0)\b', l)
        if not m:
            continue
        data = regs(m.group(1))
        for j, (i2, l2) in enumerate(ins[n + 1:n + 3]):
            op = l2.split()[0]
            if not op.startswith('v_') or op.startswith('v_cmp'):
                if op.startswith(('s_nop',)):
                    break
                continue
            dst = l2.split(None, 1)[1].split(',')[0].strip()
            if regs(dst) & data:
                print(f"{path}:{i + 1}: {l}\n{path}:{i2 + 1}:     {l2}   <- overwrites store data after {j} wait state(s)")
                found += 1
    return found


if __name__ == "__main__":
    total = sum(scan(p) for p in sys.argv[1:])
    print(f"{total} site(s)")
    sys.exit(1 if total else 0)
