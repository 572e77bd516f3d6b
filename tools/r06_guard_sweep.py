#!/usr/bin/env python3
"""Where does the two-term fp16 scheme stop being fp32-grade as a checkpoint's input-channel spread grows?  (the
10-binade limit of vqvae_weights_range_check_f32 was calibrated on two synthetic constructions at 16-19 binades only.)
tests/hetero.py's re-parametrisations of the default model at growing decades d; per d: the guard's largest spread (binades), the z_e
and x_hat error of the two-term fp16 / three-term bf16 schemes against the fp64 evaluation of the same network in units of the
(image, channel) maximum (the fp32 reference's own distance beside them), index flips against the reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import torch_port
from tests import hetero
from vqvae_amd import _lib, conv, functional as F
from vqvae_amd.modules import VQVAE

dev = torch.device("cuda:0")
conv.set_conv_backend("hip")
B = 64
sd0 = torch_port.init_state_dict(128, 32, 512, 64, seed=0, n_res_layers=2)
L = _lib.load()


def rel64(got, ref64):
    got, ref64 = np.asarray(got, np.float64), np.asarray(ref64, np.float64)
    cmax = np.maximum(np.abs(ref64).max(axis=(2, 3), keepdims=True), 1e-300)
    return float((np.abs(got - ref64) / cmax).max())


print(f"{'construction':14s} {'images':7s} {'d':>5s} {'spread':>7s} {'guard':>7s} | z_e vs fp64: {'fp32 ref':>9s} {'fp16x2':>9s} {'bf16x3':>9s} | x_hat: {'fp16x2':>9s} {'bf16x3':>9s} | flips fp16x2 / bf16x3 of {B * 64}")
for kind in ("coupled", "independent"):
    for images in ("normal", "mixed"):
        for d in (0.0, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0, 2.5, 3.0):
            sd = {k: v.clone() for k, v in sd0.items()} if d == 0 else (hetero.rescale_coupled(sd0, 1, d, 2) if kind == "coupled" else hetero.rescale_independent(sd0, 1, d, 2))
            x = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(77)) if images == "normal" else hetero.outlier_images(B, 78, "mixed")
            sd64 = {k: v.double() for k, v in sd.items()}
            with torch.no_grad():
                z_e = torch_port.encode(sd, x.clone(), 2)
                z_e64 = torch_port.encode(sd64, x.double(), 2)
                _, z_q, _, _, idx = torch_port.quantize(z_e, sd["vector_quantization.embedding.weight"], 0.25)
                x_hat64 = torch_port.decode(sd64, z_q.double(), 2)
            m = VQVAE(128, 32, 2, 512, 64, 0.25).eval()
            m.load_state_dict(sd)
            m = m.to(dev)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                flags, spreads = m.scheme_hint()
            cw, _keep = m._c_weights()
            xd = x.to(dev).contiguous()
            nws = L.vqvae_workspace_bytes(cw.dims, B, 32, 32)
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            res = {}
            for name, fl in (("fp16x2", 0), ("bf16x3", F.FWD_CONV_BF16_SPLIT)):
                with torch.no_grad():
                    ze = torch.empty(B, 8, 8, 64, device=dev)
                    _lib.check(L.vqvae_encoder_ex_f32(cw, xd.data_ptr(), B, 32, 32, fl, ze.data_ptr(), ws.data_ptr(), nws, st))
                    zq_rows = z_q.to(dev).permute(0, 2, 3, 1).contiguous()
                    xh = torch.empty_like(xd)
                    _lib.check(L.vqvae_decoder_ex_f32(cw, zq_rows.data_ptr(), B, 8, 8, fl, xh.data_ptr(), ws.data_ptr(), nws, st))
                    out = m._forward_c(xd, want_idx=True, fwd_flags=fl)
                torch.cuda.synchronize()
                res[name] = (rel64(ze.permute(0, 3, 1, 2).cpu().numpy(), z_e64.numpy()), rel64(xh.cpu().numpy(), x_hat64.numpy()),
                             int((out[3].view(-1).cpu() != idx.view(-1)).sum()))
            print(f"{kind:14s} {images:7s} {d:5.2f} {max(spreads):7.2f} {'bf16x3' if flags else 'fp16x2':>7s} |              "
                  f"{rel64(z_e.numpy(), z_e64.numpy()):9.2e} {res['fp16x2'][0]:9.2e} {res['bf16x3'][0]:9.2e} |        "
                  f"{res['fp16x2'][1]:9.2e} {res['bf16x3'][1]:9.2e} | {res['fp16x2'][2]:4d} / {res['bf16x3'][2]:4d}", flush=True)
