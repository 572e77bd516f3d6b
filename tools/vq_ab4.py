#!/usr/bin/env python3
"""Round-4 A/B of vq_track_kernel_d64 on the MODEL'S OWN z_e distribution (K=512, D=64) at 65 536 / 262 144 / 2 097 152 rows:
    rows     row-major rows (the whole path's internal layout)
    nchw     the reference's NCHW boundary layout read and written directly (round 4: 16-byte accesses + an fp32 LDS transposition)
Kernel time = HIP events around the launch (vqvae_profile_*), best and median of `iters` launches; indices and z_q of the
three forms are compared bit for bit at every size."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vqvae_amd import _lib, functional as F
from vqvae_amd.modules import VQVAE


def z_nchw(n_images, dev):
    torch.manual_seed(0)
    m = VQVAE(128, 32, 2, 512, 64, 0.25).eval().to(dev)
    outs = []
    with torch.no_grad():
        for i in range(0, n_images, 4096):
            x = torch.randn(min(4096, n_images - i), 3, 32, 32, device=dev)
            outs.append(m.pre_quantization_conv(m.encoder(x)).contiguous())          # (B, 64, 8, 8)
    return torch.cat(outs), m.vector_quantization.embedding.weight.detach().contiguous()


def time_form(z, cb, iters, rowmajor, **kw):
    ws = F.vq_workspace(cb.shape[0], cb.shape[1], z.device)
    out = F.vq_forward(z, cb, 0.25, rowmajor=rowmajor, workspace=ws, **kw)
    for _ in range(3):
        F.vq_forward(z, cb, 0.25, rowmajor=rowmajor, workspace=ws, prepared=True, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        _lib.profile_enable(True)
        F.vq_forward(z, cb, 0.25, rowmajor=rowmajor, workspace=ws, prepared=True, **kw)
        ms, n = _lib.profile_collect('vq_main')
        _lib.profile_enable(False)
        ts.append(ms / max(n, 1) * 1e3)
    return out, min(ts), statistics.median(ts)


def main():
    dev = torch.device("cuda:0")
    iters = int(os.environ.get("VQ_AB_ITERS", "30"))
    for n_images in (1024, 4096, 32768):
        zn, cb = z_nchw(n_images, dev)
        zr = zn.permute(0, 2, 3, 1).contiguous()
        N = zr.shape[0] * 64
        res = {}
        for name, z, rm, kw in (("rows", zr, True, {"form": 8}), ("rows16", zr, True, {"form": 16}), ("nchw", zn, False, {})):
            (loss, zq, ppl, idx, hist), best, med = time_form(z, cb, iters, rm, **kw)
            if not rm:
                zq = zq.permute(0, 2, 3, 1).contiguous()
            res[name] = (idx, zq, loss, best, med)
        ref = res["rows"]
        line = {"rows": N}
        for name, r in res.items():
            line[name] = {"best_us": round(r[3], 2), "median_us": round(r[4], 2), "frac_of_8TBps": round(N * 520 / r[3] / 8e6, 4),
                          "idx_equal": bool(torch.equal(r[0], ref[0])), "zq_equal": bool(torch.equal(r[1].view(torch.int32), ref[1].view(torch.int32))),
                          "loss_rel": abs(r[2].item() - ref[2].item()) / abs(ref[2].item())}
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
