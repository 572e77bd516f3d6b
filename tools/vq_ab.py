#!/usr/bin/env python3
"""A/B of the resident-image quantizer kernels on the MODEL'S OWN z_e distribution (K=512, D=64, row-major rows):
round 3's stream tracker (default) against round 2's top-3-key tracker, at 65 536 / 262 144 / 2 097 152 rows.
Kernel time = HIP events around the launch (vqvae_profile_*), best and median of `iters` launches; the two forms'
indices and z_q are compared bit for bit at every size (the top-3 form is pinned to the oracle by tests/test_vq_gpu.py)."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vqvae_amd import _lib, functional as F
from vqvae_amd.modules import VQVAE


def z_rows(n_images, dev):
    torch.manual_seed(0)
    m = VQVAE(128, 32, 2, 512, 64, 0.25).eval().to(dev)
    outs = []
    with torch.no_grad():
        for i in range(0, n_images, 4096):
            x = torch.randn(min(4096, n_images - i), 3, 32, 32, device=dev)
            outs.append(m.pre_quantization_conv(m.encoder(x)).permute(0, 2, 3, 1).contiguous())
    return torch.cat(outs), m.vector_quantization.embedding.weight.detach().contiguous()


def time_form(z, cb, iters, **kw):
    ws = F.vq_workspace(cb.shape[0], cb.shape[1], z.device)
    out = F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, **kw)
    for _ in range(3):
        F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        _lib.profile_enable(True)
        F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True, **kw)
        ms, n = _lib.profile_collect('vq_main')
        _lib.profile_enable(False)
        ts.append(ms / max(n, 1) * 1e3)
    return out, min(ts), statistics.median(ts)


def main():
    dev = torch.device("cuda:0")
    iters = int(os.environ.get("VQ_AB_ITERS", "30"))
    for n_images in (1024, 4096, 32768):
        z, cb = z_rows(n_images, dev)
        N = z.shape[0] * z.shape[1] * z.shape[2]
        res = {}
        for name, kw in (("track", {}), ("top3", {"top3_keys": True})):
            (loss, zq, ppl, idx, hist), best, med = time_form(z, cb, iters, **kw)
            res[name] = (idx, zq, loss, best, med)
        same_idx = bool(torch.equal(res["track"][0], res["top3"][0]))
        same_zq = bool(torch.equal(res["track"][1].view(torch.int32), res["top3"][1].view(torch.int32)))
        line = {"rows": N, "idx_equal": same_idx, "zq_equal": same_zq,
                "loss_rel": abs(res["track"][2].item() - res["top3"][2].item()) / abs(res["top3"][2].item())}
        for name in ("track", "top3"):
            best, med = res[name][3], res[name][4]
            line[name] = {"best_us": round(best, 2), "median_us": round(med, 2),
                          "TBps_best": round(N * 520 / best / 1e6, 3), "frac_of_8TBps": round(N * 520 / best / 8e6, 4)}
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
