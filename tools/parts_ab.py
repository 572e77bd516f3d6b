#!/usr/bin/env python3
"""Whole forward at BASELINE config 3: one vqvae_forward_f32 call against the step in n parts on n side streams
(vqvae_forward_begin / part / end), interleaved in one process: ms per step, median of `reps` repeats of `steps` steps."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd.modules import VQVAE

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = VQVAE(128, 32, 2, 512, 64, 0.25).eval().to(dev)
x = torch.randn(4096, 3, 32, 32, device=dev)
steps, reps = 30, 9
forms = [1, 2, 4, "py2", "py4"]
res = {n: [] for n in forms}
side = [torch.cuda.Stream() for _ in range(4)]


def step(n):
    if isinstance(n, int):
        return m._forward_c(x, parts=n)
    k = int(n[2:])                                     # "pyK": K independent chunk forwards on K streams (separate losses)
    cur = torch.cuda.current_stream()
    outs = []
    for i, c in enumerate(x.chunk(k)):
        side[i].wait_stream(cur)
        with torch.cuda.stream(side[i]):
            outs.append(m._forward_c(c, parts=1))
    for i in range(k):
        cur.wait_stream(side[i])
    return outs


with torch.no_grad():
    for n in forms:
        for _ in range(3):
            step(n)
    for r in range(reps):
        for n in forms:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                step(n)
            torch.cuda.synchronize()
            res[n].append((time.perf_counter() - t0) / steps * 1e3)
for n, v in res.items():
    print(f"parts {str(n):4s}: median {statistics.median(v):.4f} ms/step  min {min(v):.4f}  ({4096 / statistics.median(v) / 1e3:.3f} M images/s)")
