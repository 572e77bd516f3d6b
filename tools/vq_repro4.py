#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import c_oracle
from vqvae_amd import functional as F
dev = torch.device("cuda:0")
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vq_hetero_unit.npz"))
z, cb0 = torch.from_numpy(d["z_rows"]), torch.from_numpy(d["codebook"])
def run(zrows, cb, label, **kw):
    zz = zrows.reshape(-1, 8, 8, 64).contiguous()
    ref = c_oracle.vq_forward(zz.permute(0, 3, 1, 2).contiguous().numpy(), cb.numpy(), 0.25)["idx"].reshape(-1)
    idx = F.vq_forward(zz.to(dev), cb.to(dev), 0.25, rowmajor=True, **kw)[3].cpu().numpy().reshape(-1)
    bad = np.nonzero(idx != ref)[0]
    print(f"{label:60s}: {len(bad)} mismatches {bad[:8].tolist()} got {idx[bad[:4]].tolist()} want {ref[bad[:4]].tolist()}", flush=True)
one = torch.zeros(64, 64); one[21] = z[21]
run(one, cb0, "row 21 alone")
for kill in ([489], [68], [430], [68, 430], [464], [489, 68, 430], [297]):
    cb = cb0.clone()
    for k in kill: cb[k] = cb0[k] * 3.0 + 100.0          # far away
    run(one, cb, f"row 21 alone, codes {kill} moved far away")
for pos in (0, 5, 21, 31, 32, 40, 53, 63):
    o = torch.zeros(64, 64); o[pos] = z[21]
    run(o, cb0, f"the row at position {pos}")
o = torch.randn(64, 64, generator=torch.Generator().manual_seed(1)) * 0.01; o[21] = z[21]
run(o, cb0, "row 21 among small random rows")
# scale test: the same geometry at unit scale (divide z and cb by 64)
run(one / 64.0, cb0 / 64.0, "row 21 alone, everything / 64")
run(one * 0.001, cb0 * 0.001, "row 21 alone, everything * 1e-3")
