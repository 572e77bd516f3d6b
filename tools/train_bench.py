#!/usr/bin/env python3
"""Training-step time (forward + loss + backward, no optimizer) of main.py:74-78: convs on the HIP kernels vs
torch's conv autograd (MIOpen), the quantizer on the HIP forward/backward either way.
    train_bench.py [B] [backends] [steps] [adam]     adam: + main.py:59,80 (optim.Adam(amsgrad=True).step(); every
                                                     layer's weights are packed again each step)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd import conv, training as T, _lib
from vqvae_amd.modules import VQVAE

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
model = VQVAE(128, 32, 2, 512, 64, 0.25).to(dev).train()
x = torch.randn(B, 3, 32, 32, device=dev)
backends = sys.argv[2].split(",") if len(sys.argv) > 2 else ["hip", "torch"]
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
with_adam = len(sys.argv) > 4 and sys.argv[4] == "adam"
opt = torch.optim.Adam(model.parameters(), lr=3e-4, amsgrad=True) if with_adam else None
for backend in backends:
    conv.set_conv_backend(backend)

    def step():
        model.zero_grad(set_to_none=True)
        el, xh, pp = model(x)
        stats = T.step_losses(el, xh, pp, x, 0.06)
        stats[1].backward()
        if opt is not None:
            opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = nsteps
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"backend={backend:5s} B={B}{' +adam' if with_adam else ''}: {dt * 1e3:8.2f} ms per forward+backward   {B / dt / 1e3:8.1f} k img/s", flush=True)
conv.set_conv_backend("hip")
