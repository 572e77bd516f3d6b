#!/usr/bin/env python3
"""Hard-row probe: rows with three near-duplicate best codes in ONE lane half and three different cells (-> the stream tracker's
'hard' verdict -> rescan -> exact tasks).  Which (tile, r) placements of the true argmin come back wrong?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import c_oracle
from vqvae_amd import functional as F
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
K, D = 512, 64
def code_of(tile, r, h): return 32 * tile + (r & 3) + 8 * (r >> 2) + 4 * h
fails = []
ntrials = 0
for trial in range(40):
    cb = torch.randn(K, D, generator=g)
    rows, meta = [], []
    for n in range(64):
        h = int(torch.randint(0, 2, (1,), generator=g))
        tiles = torch.randperm(16, generator=g)[:3].tolist()
        rs = torch.randint(0, 16, (3,), generator=g).tolist()
        ks = [code_of(t, r, h) for t, r in zip(tiles, rs)]
        if len(set(ks)) < 3: continue
        base = torch.randn(D, generator=g)
        for j, k in enumerate(ks):
            cb[k] = base + 1e-4 * torch.randn(D, generator=g)
        rows.append(base + 1e-4 * torch.randn(D, generator=g)); meta.append((h, tiles, rs, ks))
    while len(rows) < 64: rows.append(torch.zeros(D)); meta.append(None)
    # only every 8th row is such a row (few tasks per unit: the task path, not the overflow path)
    z = torch.zeros(64, D)
    keep = list(range(0, 64, 8))
    for i in keep: z[i] = rows[i]
    zz = z.reshape(1, 8, 8, D).contiguous()
    ref = c_oracle.vq_forward(zz.permute(0, 3, 1, 2).contiguous().numpy(), cb.numpy(), 0.25)["idx"].reshape(-1)
    idx = F.vq_forward(zz.to(dev), cb.to(dev), 0.25, rowmajor=True)[3].cpu().numpy().reshape(-1)
    for i in keep:
        if meta[i] is None: continue
        ntrials += 1
        if idx[i] != ref[i]:
            h, tiles, rs, ks = meta[i]
            j = ks.index(int(ref[i])) if int(ref[i]) in ks else -1
            jg = ks.index(int(idx[i])) if int(idx[i]) in ks else -1
            fails.append((i, h, tiles, rs, ks, int(ref[i]), int(idx[i]), j, jg))
print(f"{len(fails)} wrong of {ntrials} hard rows")
for f in fails[:30]: print("row %d h %d tiles %s rs %s codes %s want %d got %d (want is code #%d, got #%d)" % f)
