#!/usr/bin/env python3
"""Time the conv layers of the B=4096 forward one by one through the C ABI (HIP events, 10 launches each).
"TF" counts the bf16 MFMA flop issued by the split-bf16 path (6 products per fp32 product)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vqvae_amd import _lib
if os.environ.get("VQVAE_BENCH_LIB"):          # A/B runs against another build of the library
    _lib.LIB_PATH = os.environ["VQVAE_BENCH_LIB"]
from vqvae_amd import conv_hip
from vqvae_amd.modules import ResidualLayer

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # 2 = relu_out, |4 = exact fp32 MFMA
cases = [("enc2 4x4s2 64->128", 0, nn.Conv2d(64, 128, 4, 2, 1), (B, 16, 16, 64), 64, 128, 64 * 1024 * 128),
         ("enc4 3x3 128->128", 1, nn.Conv2d(128, 128, 3, 1, 1), (B, 8, 8, 128), 128, 128, 64 * 1152 * 128),
         ("preq 1x1 128->64", 2, nn.Conv2d(128, 64, 1), (B, 8, 8, 128), 128, 64, 64 * 128 * 64),
         ("dec0 T3x3 64->128", 3, nn.ConvTranspose2d(64, 128, 3, 1, 1), (B, 8, 8, 64), 64, 128, 64 * 576 * 128),
         ("dec2 T4x4s2 128->64", 4, nn.ConvTranspose2d(128, 64, 4, 2, 1), (B, 8, 8, 128), 128, 64, 256 * 512 * 64)]


def timeit(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


tot = 0.0
for name, kind, m, shp, ci, co, macs in cases:
    m = m.to(dev)
    x = torch.randn(*shp, device=dev)
    us = timeit(lambda: conv_hip.conv(kind, x, m, m.weight, m.bias, ci, co, flags))
    tot += us
    print(f"{name:22s} {us:7.1f} us  {12 * macs * B / us / 1e6:6.0f} TF")
layer = ResidualLayer(128, 128, 32).to(dev)
x = torch.randn(B, 8, 8, 128, device=dev)
us = timeit(lambda: conv_hip.res_layer(x, layer, 2 | (flags & 4)))
print(f"{'res 3x3 128->32->128':22s} {us:7.1f} us  {12 * 64 * (1152 * 32 + 32 * 128) * B / us / 1e6:6.0f} TF")
print(f"sum of convs + 4 x res: {tot + 4 * us:.0f} us  [{os.path.basename(_lib.LIB_PATH)}]")
