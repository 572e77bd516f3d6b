import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from vqvae_amd import functional as F
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
K, D = 512, 64
cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K).to(dev)
z = (torch.randn(32, 8, 8, D, generator=g) * 0.066).to(dev)
a = F.vq_forward(z, cb, 0.25, rowmajor=True)
b = F.vq_forward(z, cb, 0.25, rowmajor=True, top3_keys=True)
torch.cuda.synchronize()
za, zb = a[1].reshape(-1, D).cpu().numpy(), b[1].reshape(-1, D).cpu().numpy()
ia, ib = a[3].reshape(-1).cpu().numpy(), b[3].reshape(-1).cpu().numpy()
print("idx equal", np.array_equal(ia, ib), "loss", a[0].item(), b[0].item())
diff = za.view(np.uint32) != zb.view(np.uint32)
rows = np.where(diff.any(1))[0]
print("rows differing", len(rows), "of", za.shape[0], rows[:40])
for r in rows[:6]:
    cols = np.where(diff[r])[0]
    print(r, "cols", cols[:16], "a", za[r, cols[:4]], "b", zb[r, cols[:4]], "z", z.reshape(-1, D)[r, cols[:4]].cpu().numpy(), "e", cb[ia[r], cols[:4]].cpu().numpy())
