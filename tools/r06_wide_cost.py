#!/usr/bin/env python3
"""What ONE row with m near-tied candidates costs the stand-alone quantizer (262 144 rows, K = 512, D = 64): all rows but one sit on
well-separated codes (closed), one row sits on a cluster of m near-identical codes.  The kernel's time is its slowest wave's, so
t(m) - t(0) is that row's exact part: m <= 64 candidates run as tasks (4 per pass), more overflow the task table and the wave takes
torch.argmin over all codes.    python tools/r06_wide_cost.py [nrows_with_cluster]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd import _lib, functional as F

dev = torch.device("cuda:0")
K, D, N = 512, 64, 262144
nhit = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator().manual_seed(0)
for m in (0, 2, 3, 8, 16, 32, 48, 64, 65, 100, 200, 450):
    cb = torch.randn(K, D, generator=g)
    if m:
        cb[:m] = cb[0] * (1 + 1e-7 * torch.randn(m, 1, generator=g)) + 1e-7 * torch.randn(m, D, generator=g)
    sel = torch.randint(m if m else 0, K, (N,), generator=g)
    z = cb[sel] + 0.05 * torch.randn(N, D, generator=g)
    if m:
        hit = torch.arange(nhit) * (N // nhit) + 5
        z[hit] = cb[0] + 1e-4 * torch.randn(nhit, D, generator=g)
    zd = z.view(N // 64, 8, 8, D).to(dev).contiguous()
    cbd = cb.to(dev)
    ws = F.vq_workspace(K, D, dev)
    out = F.vq_forward(zd, cbd, 0.25, rowmajor=True, workspace=ws)
    for _ in range(3):
        F.vq_forward(zd, cbd, 0.25, rowmajor=True, workspace=ws, prepared=True)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(20):
        F.vq_forward(zd, cbd, 0.25, rowmajor=True, workspace=ws, prepared=True)
    ms, cnt = _lib.profile_collect("vq_main")
    _lib.profile_enable(False)
    # exactness against fp64 on the cluster rows is the tests' business; here only the time
    print(f"m = {m:4d} candidates on {nhit if m else 0} row(s): {ms / cnt * 1e3:7.1f} us", flush=True)
