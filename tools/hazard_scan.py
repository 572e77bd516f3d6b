#!/usr/bin/env python3
"""Scans gfx950 assembly (hipcc -S) for a store-data hazard hipcc (ROCm 7.2) does not guard: a buffer store of more than
8 bytes whose soffset is an SGPR (or a global store with an saddr pair), followed within two wait states by an instruction
that writes one of its data registers (vector ALU results; `s_nop N` counts N + 1 wait states).  LLVM's hazard recognizer exempts the SGPR-soffset form (GCNHazardRecognizer::createsVALUHazard); on MI355X the
overwrite corrupts the last dword of the last lanes of each row (found with vq_track_kernel_d64's z_q stores, round 3).
Round 4 added a second pattern, a MISCOMPILE rather than a hazard: four __builtin_amdgcn_fdot2 calls on the components of one
loaded 16-byte vector came out as four `v_dot2c_f32_f16 vD, vS, vS` reading the SAME register vS (the first component), so a
row norm was 4 (x0^2 + x1^2) instead of the sum over eight values; the sources now use inline assembly (common.h, sqsum8_f16).
Two or more dot2c instructions with identical destination AND identical sources inside a window of four instructions, with no
write to that source in between, are reported.
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
        # 16- / 12-byte stores whose address carries an SGPR offset: the buffer form with an soffset register (the one that
        # bit us) and, as a precaution, the global form with an saddr pair
        m = re.match(r'buffer_store_dwordx[34]\s+(v\[\d+:\d+\]),\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0)\b', l) or \
            re.match(r'global_store_dwordx[34]\s+v\d+,\s*(v\[\d+:\d+\]),\s*s\[\d+:\d+\]', l)
        if not m:
            continue
        data = regs(m.group(1))
        waits = 0                                               # wait states between the store and the candidate writer
        for i2, l2 in ins[n + 1:n + 4]:
            if waits >= 2:
                break
            op = l2.split()[0]
            if op == 's_nop':
                waits += int(l2.split()[1], 0) + 1                # s_nop N = N + 1 wait states
                continue
            # (vector ALU results only.  Loads landing in the data registers were tried as "writers" too -- hundreds of sites,
            # e.g. every tile_epilogue store followed by the next ds_read_b128 into the same registers, all verified bit-exact on
            # hardware: a load returns tens of cycles after the store has read its data, the hazard window is two wait states)
            writer = op.startswith('v_') and not op.startswith('v_cmp')
            if writer and len(l2.split(None, 1)) > 1:
                dst = l2.split(None, 1)[1].split(',')[0].strip()
                if regs(dst) & data:
                    print(f"{path}:{i + 1}: {l}\n{path}:{i2 + 1}:     {l2}   <- overwrites store data after {waits} wait state(s)")
                    found += 1
                    break
            waits += 1
    # ---- repeated v_dot2c on one source register (the fdot2 miscompile)
    for n, (i, l) in enumerate(ins):
        m = re.match(r'v_dot2c_f32_f16(?:_e32)?\s+(v\d+),\s*(v\d+),\s*(v\d+)\s*$', l)
        if not m or m.group(2) != m.group(3):
            continue
        for i2, l2 in ins[n + 1:n + 4]:
            op = l2.split()[0]
            if l2 == l:
                print(f"{path}:{i + 1}: {l}\n{path}:{i2 + 1}:     {l2}   <- the same dot product twice: the fdot2 miscompile (common.h, sqsum8_f16)")
                found += 1
                break
            if op.startswith(('v_', 'ds_read', 'buffer_load', 'global_load')) and len(l2.split(None, 1)) > 1:
                dst = l2.split(None, 1)[1].split(',')[0].strip()
                if regs(dst) & regs(m.group(2)):
                    break                                       # the source was legitimately rewritten in between
    return found


if __name__ == "__main__":
    total = sum(scan(p) for p in sys.argv[1:])
    print(f"{total} site(s)")
    sys.exit(1 if total else 0)
