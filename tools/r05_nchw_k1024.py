"""Round 5: the module's NCHW boundary with a K = 1024 codebook -- vq_track_kernel_d64<4, true, 1> (the 32 x 64 block turned
around in two halves) beside the two-sweep filter kernel it replaces for that layout, and beside the row-major four-wave form.
    python tools/r05_nchw_k1024.py  > gpurun_out/r05/nchw_k1024.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd import _lib, functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for (B, H, K) in ((512, 56, 1024), (4096, 8, 1024), (64, 56, 1024), (512, 56, 768)):
    cb = torch.empty(K, 64, device=dev).uniform_(-1 / K, 1 / K)
    z = torch.randn(B, 64, H, H, device=dev) * 0.02
    zr = z.permute(0, 2, 3, 1).contiguous()
    ws = F.vq_workspace(K, 64, dev)
    F.vq_forward(z, cb, 0.25, workspace=ws)
    n = B * H * H
    out = {}
    for name, fn in (("nchw track", lambda: F.vq_forward(z, cb, 0.25, workspace=ws, prepared=True)),
                     ("nchw filter", lambda: F.vq_forward(z, cb, 0.25, workspace=ws, prepared=True, bf16_filter=True)),
                     ("rows track", lambda: F.vq_forward(zr, cb, 0.25, rowmajor=True, workspace=ws, prepared=True))):
        med, best = timed(fn)
        out[name] = med
        print(f"B={B} {H}x{H} K={K} rows={n}: {name:12s} median {med:8.1f} us  best {best:8.1f} us  "
              f"{n * 520 / med / 1e6 / 8:.3f} of 8 TB/s   [{_lib.vq_kernel_instance(n, K, 64, H * H, 0 if name[0] == 'n' else 1)}]")
    a = F.vq_forward(z, cb, 0.25, workspace=ws, prepared=True)
    b = F.vq_forward(z, cb, 0.25, workspace=ws, prepared=True, bf16_filter=True)
    torch.cuda.synchronize()
    assert torch.equal(a[3], b[3]) and torch.equal(a[1], b[1]) and torch.equal(a[4], b[4])
    print(f"   identical bits; nchw track / nchw filter = {out['nchw filter'] / out['nchw track']:.2f} x faster")
