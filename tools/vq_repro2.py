#!/usr/bin/env python3
"""Narrowing the hetero repro: the failing unit alone, the failing row alone (repeated), subsets of the unit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import c_oracle
from vqvae_amd import functional as F
dev = torch.device("cuda:0")
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vq_hetero_unit.npz"))
z, cb = torch.from_numpy(d["z_rows"]), torch.from_numpy(d["codebook"])
r, r0 = 21, 0                     # the failing row inside the unit (global row 2837 of the test batch)
def run(rows, label, **kw):
    zz = rows.reshape(-1, 8, 8, 64).contiguous()          # (B, 8, 8, 64) row-major
    ref = c_oracle.vq_forward(zz.permute(0, 3, 1, 2).contiguous().numpy(), cb.numpy(), 0.25)["idx"].reshape(-1)
    idx = F.vq_forward(zz.to(dev), cb.to(dev), 0.25, rowmajor=True, **kw)[3].cpu().numpy().reshape(-1)
    bad = np.nonzero(idx != ref)[0]
    print(f"{label:50s}: {len(bad)} mismatches {bad[:8].tolist()} got {idx[bad[:4]].tolist()} want {ref[bad[:4]].tolist()}", flush=True)
unit = z[r0:r0 + 64]
run(unit, "the unit alone")
run(z[r:r + 1].repeat(64, 1), "row 21 x 64")
one = torch.zeros(64, 64); one[21] = z[r]
run(one, "row 2837 at position 21, other rows zero")
hard = [10, 18, 19, 21]
only_hard = torch.zeros(64, 64)
for h in hard: only_hard[h - r0] = z[h]
run(only_hard, "the four hard rows, others zero")
for h in hard:
    u = unit.clone(); u[h - r0] = 0
    run(u, f"unit without row {h}")
u = unit.clone()
for h in hard:
    if h != 21: u[h - r0] = 0
run(u, "unit with 21 as the only hard row")
closed_only = unit.clone()
run(torch.cat([unit, unit]), "unit twice (128 rows)")
