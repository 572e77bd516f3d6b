import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import c_oracle
from vqvae_amd import functional as F
K,D,B,H,W=512,64,33,8,8
g = torch.Generator().manual_seed(K * 7 + D + B)
cb = torch.randn(K, D, generator=g); z = torch.randn(B, D, H, W, generator=g)
ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
dev = torch.device("cuda:0")
zr = z.permute(0, 2, 3, 1).contiguous().to(dev)
loss, zq, ppl, idx, hist = F.vq_forward(zr, cb.to(dev), 0.25, rowmajor=True)
dbg = zq.cpu().numpy().reshape(-1, D)[:, :8]
idx = idx.cpu().numpy().reshape(-1); ri = ref["idx"].reshape(-1)
print("mismatch rows", np.nonzero(idx != ri)[0])
for n in (899, 0, 1, 898, 900):
    print(n, "v1 %.3f v2 %.3f v3 %.3f c1 %d c2 %d delta %.3f flags %d zn %.4f | got %d want %d" % (*dbg[n][:3], int(dbg[n][3]), int(dbg[n][4]), dbg[n][5], int(dbg[n][6]), dbg[n][7], idx[n], ri[n]))
fl = dbg[:,6].astype(int)
print("pair rows", (fl&1).sum(), "hard", ((fl>>1)&1).sum(), "bad", ((fl>>2)&1).sum())
zf = zr.cpu().numpy().reshape(-1, D).astype(np.float64); e = cb.numpy().astype(np.float64)
A = 2.0**11
s = (e @ zf[899] - 0.5*(e**2).sum(1))*A
o = np.argsort(-s)[:4]; print("true scaled scores top4", o, s[o])
