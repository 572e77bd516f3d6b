#!/usr/bin/env python3
"""Fused VQ kernel at BASELINE config-3 size: kernel time, agreement with the exhaustive kernel and -- for a
-DVQ_TIMING build (tools/build_variant.py) -- the per-phase wall-clock breakdown averaged over all waves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd import _lib
if os.environ.get("VQVAE_BENCH_LIB"):
    _lib.LIB_PATH = os.environ["VQVAE_BENCH_LIB"]
from vqvae_amd import functional as F

dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
K, D = 512, 64
N = int(os.environ.get("VQ_ROWS", 262144))
cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K).to(dev)
z = (torch.randn(N // 64, 8, 8, D, generator=g) * 0.066).to(dev)
ws = F.vq_workspace(K, D, dev)
out = F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws)
ref = F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=F.vq_workspace(K, D, dev), exact_sweep=True)
same = all(torch.equal(a, b) for a, b in zip(out, ref) if isinstance(a, torch.Tensor))
for _ in range(3):
    F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True)
_lib.profile_enable(True)
for _ in range(30):
    F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True)
kms, kn = _lib.profile_collect('vq_main')
_lib.profile_enable(False)
us = kms / max(kn, 1) * 1e3
print(f"[{os.path.basename(_lib.LIB_PATH)}] N={N} kernel {us:.2f} us  {N * 520 / us / 1e6:.3f} TB/s  frac {N * 520 / us / 1e6 / 8:.3f}  "
      f"outputs == exhaustive kernel: {same}")
if os.environ.get("VQ_TIMING") == "span":
    # -DVQ_SWEEP_TIMING=2 build: per-workgroup [start, end] on the chip-wide 100 MHz clock
    off = 256
    off = (off + K * 4 + 255) // 256 * 256
    off = (off + K * D * 4 + 255) // 256 * 256 + 512 * 8
    slots = ws[off:off + 256 * 16].view(torch.int64)
    slots.zero_()
    F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True)
    torch.cuda.synchronize()
    t = slots.view(256, 2).cpu().double() * 0.01
    t0 = t[:, 0].min()
    st, en = t[:, 0] - t0, t[:, 1] - t0
    print(f"   workgroup starts: 0 .. {st.max():.2f} us (mean {st.mean():.2f});  ends: {en.min():.2f} .. {en.max():.2f} us (mean {en.mean():.2f});"
          f"  durations: mean {(en - st).mean():.2f}, min {(en - st).min():.2f}, max {(en - st).max():.2f} us")
elif os.environ.get("VQ_TIMING") == "2":
    off = 256
    off = (off + K * 4 + 255) // 256 * 256
    off = (off + K * D * 4 + 255) // 256 * 256 + 512 * 8
    slots = ws[off:off + 64 * 64].view(torch.int64)
    slots.zero_()
    F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True)
    torch.cuda.synchronize()
    t = slots.view(64, 8).cpu()
    cyc, wall = t[:, 0].double(), t[:, 1].double()
    print(f"   shader clock over the kernel: {(cyc / (wall * 10e-9)).mean() / 1e9:.3f} GHz  (wave 0 of 64 workgroups; "
          f"{wall.mean() * 0.01:.1f} us per wave)")
elif os.environ.get("VQ_TIMING") == "pc":
    off = 256
    off = (off + K * 4 + 255) // 256 * 256
    off = (off + K * D * 4 + 255) // 256 * 256 + 512 * 8
    slots = ws[off:off + 64 * 64].view(torch.int64)
    slots.zero_()
    F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True)
    torch.cuda.synchronize()
    names = ["prologue (all 8 waves, /2)", "sweeper: waiting for tiles", "sweeper: sweep + merge + publish", "I/O: convert + publish",
             "I/O: waiting for the verdict", "I/O: (rest of) gathers + exact part", "I/O: epilogue", "I/O: rows landing + loop", "exit",
             "  I/O: verdict read + tile-0 gathers issued", "  I/O: direct task list", "  I/O: rescan", "  I/O: row copy", "  I/O: task passes", "-", "-"]
    t = slots.view(32, 16).sum(0).cpu().tolist()
    for n, v in zip(names, t):
        print(f"   {n:46s} {v / 128 * 0.01:8.2f} us per wave of that role (sum over its 4 pairs, first 32 workgroups)")
elif os.environ.get("VQ_TIMING"):
    off = 256
    off = (off + K * 4 + 255) // 256 * 256
    off = (off + K * D * 4 + 255) // 256 * 256 + 512 * 8
    slots = ws[off:off + 64 * 64].view(torch.int64)
    slots.zero_()
    F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True)
    torch.cuda.synchronize()
    names = ["prologue (codebook copy, first requests)", "rows landed + fp16 conversion", "sweep", "merge + classify", "exact part",
             "epilogue", "loop exit", "-"]
    if os.environ.get("VQ_TIMING") == "pc":
        names = ["prologue (all 8 waves; x2 roles)", "sweeper: waiting for tiles", "sweeper: sweep + merge + publish", "I/O: convert + publish",
                 "I/O: waiting for the verdict", "I/O: gathers + exact part", "I/O: epilogue", "I/O: rows landing + loop"]
    if os.environ.get("VQ_TIMING") == "old":
        names = ["prologue (codebook copy, first requests)", "rows landed", "convert + sweep 1", "sweep 2", "exact part",
                 "epilogue", "loop exit", "tail: loss partial + histogram flush (x8)"]
    per_block = slots.view(64, 8).cpu()
    mx = per_block[:, 7].double() * 0.01
    print(f"   slowest wave of a workgroup: mean {mx.mean():.2f} us, max {mx.max():.2f} us, min {mx.min():.2f} us (first 64 workgroups)")
    t = per_block.sum(0).tolist()
    t[7] = 0
    for n, v in zip(names, t):
        print(f"   {n:42s} {v / (256 if os.environ.get('VQ_TIMING') == 'pc' else 512) * 0.01:8.2f} us per wave (sum over its blocks, first 64 workgroups)")
    print(f"   {'total':42s} {sum(t) / 512 * 0.01:8.2f} us")
