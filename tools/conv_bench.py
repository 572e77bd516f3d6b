#!/usr/bin/env python3
"""Time one conv layer (enc conv 3x3 128->128 at B=4096, 8x8) through the C ABI: split-bf16 vs exact fp32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from vqvae_amd import conv_hip
dev = torch.device("cuda:0"); torch.manual_seed(0)
m = nn.Conv2d(128, 128, 3, 1, 1).to(dev)
x = torch.randn(4096, 8, 8, 128, device=dev)
for name, fl in (("split-bf16", 2), ("exact-fp32", 2 | 4)):
    for _ in range(3): y = conv_hip.conv(1, x, m, m.weight, m.bias, 128, 128, fl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): conv_hip.conv(1, x, m, m.weight, m.bias, 128, 128, fl)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: conv3x3 128->128 B=4096 8x8: {ms*1e3:.1f} us  {2*262144*1152*128/ms/1e9:.1f} TFLOP/s-equivalent")
