import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vqvae_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "build", "variants", "libvqvae_dbg.so")
from vqvae_amd import functional as F
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
K, D, N = 512, 64, 65536
cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K).to(dev)
z = (torch.randn(N // 64, 8, 8, D, generator=g) * 0.066).to(dev)
loss, zq, ppl, idx, hist = F.vq_forward(z, cb, 0.25, rowmajor=True)
d = zq.cpu().numpy().reshape(-1, D)[:, :8]
fl = d[:, 6].astype(int)
print("rows", N, "pair %.4f hard %.4f bad %.4f" % ((fl & 1).mean(), ((fl >> 1) & 1).mean(), ((fl >> 2) & 1).mean()))
gap12 = d[:, 0] - d[:, 1]; gap13 = d[:, 0] - d[:, 2]
print("delta mean %.4g  |v1| mean %.4g  gap12 median %.4g  gap13 median %.4g  zn mean %.4g" % (d[:, 5].mean(), np.abs(d[:, 0]).mean(), np.median(gap12), np.median(gap13), d[:, 7].mean()))
print("frac gap12<delta %.4f  gap13<delta %.4f" % ((gap12 < d[:, 5]).mean(), (gap13 < d[:, 5]).mean()))
