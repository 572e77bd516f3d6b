#!/usr/bin/env python3
"""Winograd F(2x2, 3x3) for the encoder's 3x3 128 -> 128 layer under the two-term fp16 product scheme: what it would do to the
error budget and to the work around the MFMAs (a keep / kill decision with numbers, not a half-landed kernel).
CPU only; the products are emulated exactly in fp64 from the fp16 terms (as tests/test_fp16_scheme_cpu.py does), so what is
measured is the scheme's representation error through the transforms, not an accumulation order.

  direct    x = h1 + h2 per image scale, w = g1 + g2 per output-channel scale, y = sum of the three term products over 9 taps
  winograd  V = B^T d B per 4x4 input tile (entries are sums of four activations: computed in fp32 like the kernel would, then
            split into two fp16 terms on ONE power-of-two scale per image), U = G g G^T per filter in fp64 (offline), split on
            one scale per output channel (and, variant 2, one per output channel and transformed position), M = U . V per
            position with the three term products, y = A^T M A in fp32.
Error = max over a channel of |y - exact| / max|exact channel|; the parity suite's tolerance there is 1e-5 (+ rtol 1e-4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import torch_port
from tests import hetero

E = "encoder.conv_stack."
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def split(x, scale):
    xs = (x * scale).float()
    h1 = xs.half()
    h2 = (xs - h1.float()).half()
    return h1.double() / scale, h2.double() / scale


def p2(m):
    return 2.0 ** (14 - torch.floor(torch.log2(m.clamp_min(1e-300))))


def run(sd, label):
    x = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        t = F.relu(F.conv2d(x, sd[E + "0.weight"], sd[E + "0.bias"], 2, 1))
        t = F.relu(F.conv2d(t, sd[E + "2.weight"], sd[E + "2.bias"], 2, 1)).double()          # (8, 128, 8, 8)
        w = sd[E + "4.weight"].double()                                                     # (128, 128, 3, 3)
        exact = F.conv2d(t, w, None, 1, 1)
        cmax = exact.abs().amax(dim=(0, 2, 3), keepdim=True)
        err = lambda y: float(((y - exact).abs() / cmax).max())
        fp32 = err(F.conv2d(t.float(), w.float(), None, 1, 1).double())
        # direct two-term fp16
        x1, x2 = split(t, p2(t.abs().amax(dim=(1, 2, 3), keepdim=True)))
        w1, w2 = split(w, p2(w.abs().amax(dim=(1, 2, 3), keepdim=True)))
        direct = err(F.conv2d(x1, w1, None, 1, 1) + F.conv2d(x1, w2, None, 1, 1) + F.conv2d(x2, w1, None, 1, 1))
        # winograd: tiles of 4x4 inputs at stride 2 over the zero-padded 10x10 map -> 4x4 tiles of 2x2 outputs
        tp = F.pad(t, (1, 1, 1, 1))
        d = tp.unfold(2, 4, 2).unfold(3, 4, 2)                                              # (B, C, 4, 4, 4, 4) [ty][tx][i][j]
        V = torch.einsum("ai,bctuij,kj->bctuak", BT.float(), d.float(), BT.float()).double()   # the kernel's fp32 transform
        U = torch.einsum("ai,ocij,kj->ocak", G, w, G)                                       # (O, C, 4, 4)
        res = {}
        for variant in (1, 2):
            sv = p2(V.abs().amax(dim=(1, 2, 3, 4, 5), keepdim=True))
            su = p2(U.abs().amax(dim=(1, 2, 3), keepdim=True)) if variant == 1 else p2(U.abs().amax(dim=1, keepdim=True))
            V1, V2 = split(V, sv)
            U1, U2 = split(U, su)
            M = sum(torch.einsum("ocak,bctuak->botuak", a, b) for a, b in ((U1, V1), (U1, V2), (U2, V1)))
            Y = torch.einsum("pa,botuak,qk->botupq", AT.float(), M.float(), AT.float()).double()   # fp32 output transform
            y = Y.permute(0, 1, 2, 4, 3, 5).reshape(exact.shape)
            res[variant] = err(y)
    print(f"{label:28s} fp32 conv {fp32:.2e}   direct two-term fp16 {direct:.2e}   winograd (scale per channel) {res[1]:.2e}   "
          f"(per channel and position) {res[2]:.2e}")


sd0 = torch_port.init_state_dict()
run(sd0, "default init")
for seed in (1, 2):
    run(hetero.rescale_coupled(sd0, seed), f"trained-like coupled, seed {seed}")
    run(hetero.rescale_independent(sd0, seed), f"independent factors, seed {seed}")
print("""
operand traffic per image, 3x3 128 -> 128 on an 8x8 map (two fp16 terms):
  direct    9 taps x 128 x 128 x 4 B = 589 824 B of weights per workgroup pass, 2 x 64 x 128 x 2 B = 32 KiB of activations read 9 x
            from LDS, 1 728 MFMAs (32x32x16) per image
  winograd 16 positions x 128 x 128 x 4 B = 1 048 576 B of weights (1.78 x), 16 tiles x 16 positions x 128 ch x 4 B = 128 KiB of
            transformed activations (4 x the map: tiles overlap), 768 MFMAs (2.25 x fewer) -- M = 16 tiles per image, so a
            32-row MFMA tile needs two images per wave or the 16x16x32 shape (twice the operand reads per flop)""")
