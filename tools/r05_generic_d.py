"""Round 5: what the exact-fp32 vector quantizer (vq_generic_kernel: any embedding width up to 256) costs beside the matrix-core kernels.
    python tools/r05_generic_d.py > gpurun_out/r05/generic_d.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd import _lib, functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (N, K, D) in ((262144, 512, 48), (262144, 512, 64), (65536, 512, 48), (262144, 512, 7), (262144, 1024, 96), (262144, 512, 200)):
    cb = torch.empty(K, D, device=dev).uniform_(-1 / K, 1 / K)
    z = torch.randn(N // 64, 8, 8, D, device=dev) * 0.07
    ws = F.vq_workspace(K, D, dev)
    fn = lambda: F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws)   # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"rows={N} K={K} D={D}: {_lib.vq_kernel_name(K, D):24s} median {ts[5]:9.1f} us (whole call: codebook terms + kernel + finalize)   "
          f"{N * K * D * 2 / ts[5] / 1e6:8.1f} TFLOP/s of distance arithmetic")
