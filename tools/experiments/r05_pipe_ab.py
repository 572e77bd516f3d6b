"""Round 5: the default shapes' step as ONE launch (c3_pipeline_kernel) against the four separate launches (VQVAE_FWD_NO_PIPELINE),
interleaved in one process on the same model and batch; outputs compared bit for bit first.
    python tools/experiments/r05_pipe_ab.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vqvae_amd import conv, functional as F  # noqa: E402
from vqvae_amd.modules import VQVAE  # noqa: E402

conv.set_conv_backend("hip")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = VQVAE(128, 32, 2, 512, 64, 0.25).to(dev).eval()
for B in [int(a) for a in sys.argv[1:]] or [4096, 1024, 256]:
    x = torch.randn(B, 3, 32, 32, device=dev)
    with torch.no_grad():
        a = [t.clone() for t in m._forward_c(x, want_idx=True)]
        b = [t.clone() for t in m._forward_c(x, want_idx=True, vq_flags=F.FWD_NO_PIPELINE)]
        torch.cuda.synchronize()
        same = all(torch.equal(u, v) for u, v in zip(a, b))
        res = {0: [], F.FWD_NO_PIPELINE: []}
        for rnd in range(7):
            for fl in (0, F.FWD_NO_PIPELINE):
                for _ in range(3):
                    m._forward_c(x, vq_flags=fl)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    m._forward_c(x, vq_flags=fl)
                e1.record()
                e1.synchronize()
                res[fl].append(e0.elapsed_time(e1) / 30)
        # the pipeline's wait statistics of the last call: control words behind the loss partials in the z_e region
        from vqvae_amd import _lib
        m._forward_c(x)
        torch.cuda.synchronize()
        L, (cw, _keep) = _lib.load(), m._c_weights()
        ws, _ = m._c_workspace(L, cw, B, 32, 32, dev)
        G = (B + 3) // 4
        o = L.vqvae_workspace_ze_offset(cw.dims, B, 32, 32) + ((G * 8 + 255) // 256) * 256
        c = ws[o:o + 16].view(torch.int32).cpu()
        stats = f"  tickets {int(c[0])}, polls {int(c[1])}, workgroups that waited {int(c[2])}"
    p, q = sorted(res[0]), sorted(res[F.FWD_NO_PIPELINE])
    print(f"B={B}: identical bits {same};  one launch: median {p[3]:.4f} ms (min {p[0]:.4f})   four launches: median {q[3]:.4f} ms (min {q[0]:.4f})   "
          f"ratio {q[3] / p[3]:.3f}{stats}")
