"""Round 6: the quantizer kernels behind the module boundary (NCHW maps) and the row-major entry on a TRAINED codebook against the
default init -- the two-sweep filter kernel (maps whose pixel count is not a multiple of 32) had the same one-lane overflow fallback as
the stream-tracker kernel: 1 796 us -> 122 us for 200 704 rows (vq_wave_argmin)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
from tests import cases
from vqvae_amd import _lib, functional as F
dev = torch.device("cuda:0")
name = "trained_b128x20k"
cb = cases.trained_state(name)["vector_quantization.embedding.weight"].to(dev)
g = np.load(os.path.join(ROOT, "tests/golden/trained_cases.npz"))
ze = torch.from_numpy(g[f"{name}/z_e"])          # (32,64,8,8)
rows = ze.permute(0, 2, 3, 1).reshape(-1, 64)
rows = rows.repeat(98, 1)[:1024 * 196]            # 200704 rows
for label, z in (("NCHW 14x14 maps (vq_filter)", rows.view(1024, 14, 14, 64).permute(0, 3, 1, 2).contiguous()),
                 ("NCHW 8x8 maps", rows[:196608].view(3072, 8, 8, 64).permute(0, 3, 1, 2).contiguous()),
                 ("rows", rows.view(1024, 14, 14, 64).contiguous())):
    zd = z.to(dev)
    rm = label == "rows"
    for cbk, cl in ((cb, "trained codebook"), (torch.empty(512, 64, device=dev).uniform_(-1/512, 1/512), "init codebook")):
        ws = F.vq_workspace(512, 64, dev)
        F.vq_forward(zd, cbk, 0.25, rowmajor=rm, workspace=ws)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(5):
            F.vq_forward(zd, cbk, 0.25, rowmajor=rm, workspace=ws, prepared=True)
        ms, cnt = _lib.profile_collect("vq_main")
        _lib.profile_enable(False)
        kern = "vq_filter_kernel_d64" if "14x14" in label else _lib.vq_kernel_name(512, 64, 1 if rm else 0)
        print(f"{label:30s} {cl:18s} {kern:24s} {ms / cnt * 1e3:9.1f} us", flush=True)
