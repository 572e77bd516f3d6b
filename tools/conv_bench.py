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
from vqvae_amd.modules import Decoder, Encoder
enc, dec = Encoder(3, 128, 2, 32).to(dev), Decoder(64, 128, 2, 32).to(dev)
L = _lib.load()
ximg = torch.randn(B, 3, 32, 32, device=dev)
c0 = enc.conv_stack[0]
p0 = conv_hip._packed(c0, ("conv_in",), c0.weight, lambda: L.vqvae_conv_in_packed_bytes(3, 64),
                      lambda w, buf: L.vqvae_conv_in_pack_f32(w.data_ptr(), 3, 64, buf.data_ptr(), None))
y0 = torch.empty(B, 16, 16, 64, device=dev)
us = timeit(lambda: _lib.check(L.vqvae_conv_in_forward_f32(ximg.data_ptr(), p0.data_ptr(), c0.bias.data_ptr(), B, 32, 32, 3, 64, 2,
                                                           y0.data_ptr(), torch.cuda.current_stream().cuda_stream)))
tot += us
print(f"{'conv_in 4x4s2 3->64':22s} {us:7.1f} us  {(B * 256 * 64 * 4 + B * 3072 * 4) / us / 1e3:6.0f} GB/s")
d4 = dec.inverse_conv_stack[4]
p4 = conv_hip._packed(d4, ("convt_out",), d4.weight, lambda: L.vqvae_convt_out_packed_bytes(64, 3),
                      lambda w, buf: L.vqvae_convt_out_pack_f32(w.data_ptr(), 64, 3, buf.data_ptr(), None))
xh = torch.empty(B, 3, 32, 32, device=dev)
us = timeit(lambda: _lib.check(L.vqvae_convt_out_forward_f32(y0.data_ptr(), p4.data_ptr(), d4.bias.data_ptr(), B, 16, 16, 64, 3, 0,
                                                             xh.data_ptr(), torch.cuda.current_stream().cuda_stream)))
tot += us
print(f"{'convT_out 4x4s2 64->3':22s} {us:7.1f} us  {(B * 256 * 64 * 4 + B * 3072 * 4) / us / 1e3:6.0f} GB/s")
layer = ResidualLayer(128, 128, 32).to(dev)
x = torch.randn(B, 8, 8, 128, device=dev)
us = timeit(lambda: conv_hip.res_layer(x, layer, 2 | (flags & 4)))
print(f"{'res 3x3 128->32->128':22s} {us:7.1f} us  {12 * 64 * (1152 * 32 + 32 * 128) * B / us / 1e6:6.0f} TF")
print(f"sum of convs + 4 x res: {tot + 4 * us:.0f} us  [{os.path.basename(_lib.LIB_PATH)}]")
