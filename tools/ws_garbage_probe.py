#!/usr/bin/env python3
"""Does any whole-path entry point read workspace bytes it did not write?  The same encoder / decoder / forward call with the
caller's workspace pre-filled with different byte patterns must give bitwise identical results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import torch_port
from tests import hetero
from vqvae_amd import _lib, conv
from vqvae_amd.modules import VQVAE
dev = torch.device("cuda:0")
L = _lib.load()
conv.set_conv_backend("hip")
sd0 = torch_port.init_state_dict()
for wname, sd in (("default", sd0), ("coupled2", hetero.rescale_coupled(sd0, 2))):
    m = VQVAE(128, 32, 2, 512, 64, 0.25).eval(); m.load_state_dict(sd); m = m.to(dev)
    for iname, x in (("normal", torch.randn(64, 3, 32, 32, generator=torch.Generator().manual_seed(77))), ("mixed", hetero.outlier_images(64, 78, "mixed"))):
        xd = x.to(dev).contiguous(); B = 64
        cw, _keep = m._c_weights()
        nws = L.vqvae_workspace_bytes(cw.dims, B, 32, 32)
        st = torch.cuda.current_stream().cuda_stream
        outs = []
        for fill in (0x00, 0xFF, 0x7F, 0x3C, None):
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)
            if fill is None: ws.copy_(torch.randint(0, 256, (nws,), dtype=torch.uint8, device=dev))
            else: ws.fill_(fill)
            z_e = torch.empty(B, 8, 8, 64, device=dev)
            _lib.check(L.vqvae_encoder_ex_f32(cw, xd.data_ptr(), B, 32, 32, 0, z_e.data_ptr(), ws.data_ptr(), nws, st))
            zq = torch.randn(B, 8, 8, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
            xh = torch.empty_like(xd)
            if fill is None: ws.copy_(torch.randint(0, 256, (nws,), dtype=torch.uint8, device=dev))
            else: ws.fill_(fill)
            _lib.check(L.vqvae_decoder_ex_f32(cw, zq.data_ptr(), B, 8, 8, 0, xh.data_ptr(), ws.data_ptr(), nws, st))
            torch.cuda.synchronize()
            outs.append((z_e.clone(), xh.clone()))
        same_e = [bool(torch.equal(outs[0][0].view(torch.int32), o[0].view(torch.int32))) for o in outs]
        same_d = [bool(torch.equal(outs[0][1].view(torch.int32), o[1].view(torch.int32))) for o in outs]
        print(wname, iname, "encoder equal across fills:", same_e, "decoder:", same_d, flush=True)
