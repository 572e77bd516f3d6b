#!/usr/bin/env python3
"""FETCH_SIZE calibration: a plain 1 GiB device copy (torch clone = wide coalesced float4 reads)."""
import torch
z = torch.randn(1 << 28, device="cuda:0")     # 1 GiB fp32
for _ in range(3):
    y = z.clone()
torch.cuda.synchronize()
print("copied bytes per launch", z.numel() * 4)
