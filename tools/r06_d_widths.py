"""Round 6: the stand-alone quantizer for every embedding width class at K = 512, 262 144 rows (main.py:21 leaves --embedding_dim free):
which kernel runs, its time by the dispatch's own events, fraction of 8 TB/s on the algorithmic (8 D + 8) bytes per row."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd import _lib, functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
N = 262144
for (K, D) in ((512, 32), (512, 48), (512, 64), (512, 96), (512, 128), (512, 200), (512, 256), (1024, 128)):
    cb = torch.empty(K, D, device=dev).uniform_(-1 / K, 1 / K)
    z = torch.randn(N // 64, 8, 8, D, device=dev) * 0.07
    ws = F.vq_workspace(K, D, dev)
    F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws)
    for _ in range(2):
        F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(10):
        F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True)
    ms, cnt = _lib.profile_collect("vq_main")
    _lib.profile_enable(False)
    t = ms / cnt * 1e-3
    print(f"K={K:5d} D={D:4d}: {_lib.vq_kernel_name(K, D):26s} {t * 1e6:9.1f} us/launch ({cnt // 10} launch(es) per call)  "
          f"{N * (8 * D + 8) / t / 8e12:6.3f} of 8 TB/s   {2.0 * N * K * D / t / 1e12:7.1f} TFLOP/s of distance arithmetic", flush=True)
