#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vqvae_amd import functional as F
dev = torch.device("cuda:0")
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vq_hetero_unit.npz"))
z, cb = torch.from_numpy(d["z_rows"]), torch.from_numpy(d["codebook"])
zz = z.reshape(1, 8, 8, 64).contiguous()
idx = F.vq_forward(zz.to(dev), cb.to(dev), 0.25, rowmajor=True)[3].cpu().numpy().reshape(-1)
for r in range(64):
    v = int(idx[r]); k = v & 0xFFFFF
    thr = np.array([(v >> 32) & 0xFFFFFFFF], np.uint32).view(np.float32)[0]
    fl = (v >> 20) & 15
    if True:
        print(r, "k", k & 0x3FF, "want", int(d["idx"][r]), "open", fl & 1, "hard", (fl >> 1) & 1, "bad", (fl >> 2) & 1, "hi", float(thr), "true |z^|^2", float((z[r].half().float() ** 2).sum()))
