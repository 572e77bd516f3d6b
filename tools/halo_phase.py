#!/usr/bin/env python3
"""Per-phase cycle breakdown of conv_halo8_h2_kernel from a build with phase stamps (tools/conv_stamp_patch.py;
VQVAE_BENCH_LIB=<that library>): the whole forward at BASELINE config 5's shape, batch and image size from the command line."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd import _lib
if os.environ.get("VQVAE_BENCH_LIB"):
    _lib.LIB_PATH = os.environ["VQVAE_BENCH_LIB"]
from vqvae_amd import conv
from vqvae_amd.modules import VQVAE

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
conv.set_conv_backend("hip")
m = VQVAE(128, 32, 2, 8192, 128, 0.25).eval().to(dev)
x = torch.randn(B, 3, S, S, device=dev)
raw = ctypes.CDLL(_lib.LIB_PATH)
have = hasattr(raw, "vqvae_debug_conv_stamps")
names = ["prologue", "stage", "ldA0+dma wait", "barrier", "dma issue", "raw issue", "mfma", "epilogue"]
with torch.no_grad():
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    if have:
        raw.vqvae_debug_conv_stamps(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        m(x)
    e1.record()
    torch.cuda.synchronize()
print(f"forward B={B} {S}x{S}: {e0.elapsed_time(e1) / 3:.2f} ms")
if have:
    raw.vqvae_debug_conv_stamps(buf, 0)
    for v, vn in enumerate(["enc2 4x4s2 (S2D)", "enc4 + dec0 3x3", "1x1 (no halo)", "dec2 T4x4s2 128->64"]):
        n = max(buf[16 * v + 8], 1)
        tot = sum(buf[16 * v + i] for i in range(8))
        print(f"{vn:22s} per wave (cycles): " + "  ".join(f"{names[i]} {buf[16 * v + i] / n:.0f} ({100.0 * buf[16 * v + i] / max(tot, 1):.1f}%)" for i in range(8)) + f"   total {tot / n:.0f}")
