#!/usr/bin/env python3
"""What differs between a box where the config-3 step takes 0.92 ms and one where it takes 1.08 ms (both seen in round 4 with the
same library and a bare-MFMA calibration of ~1 720 TF)?  One JSON line per run: the step, the four conv kernels and the quantizer by
the library's events, the bare-MFMA calibration, a 1 GiB device copy, and rocm-smi's power cap / clocks / temperature."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vqvae_amd import _lib, conv as conv_mod
from vqvae_amd.modules import VQVAE

dev = torch.device("cuda:0")
out = {}
out["calibration"] = {k: v for k, v in bench.calibrate(torch, dev).items() if k in ("mfma_fp16_random_tflops", "sclk_ghz")}
conv_mod.set_conv_backend("hip")
torch.manual_seed(0)
m = VQVAE(128, 32, 2, 512, 64, 0.25).eval().to(dev)
x = torch.randn(4096, 3, 32, 32, device=dev)
with torch.no_grad():
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(30):
            m(x)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
out["ms_per_step"] = round(sorted(ts)[3], 4)
out["ms_per_step_min_max"] = [round(min(ts), 4), round(max(ts), 4)]
a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
for _ in range(2):
    b.copy_(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    b.copy_(a)
torch.cuda.synchronize()
out["copy_GBps"] = round(2 * a.numel() * 4 * 10 / (time.perf_counter() - t0) / 1e9, 1)
del a, b
# the same step again right behind the copy (thermal / clock state)
with torch.no_grad():
    t0 = time.perf_counter()
    for _ in range(60):
        m(x)
    torch.cuda.synchronize()
out["ms_per_step_again"] = round((time.perf_counter() - t0) / 60 * 1e3, 4)
out["calibration_again"] = {k: v for k, v in bench.calibrate(torch, dev).items() if k in ("mfma_fp16_random_tflops", "sclk_ghz")}
try:
    smi = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks", "--showtemp", "--showperflevel", "--json"],
                         capture_output=True, text=True, timeout=30).stdout
    j = json.loads(smi)
    card = next(iter(j.values()))
    out["smi"] = {k: v for k, v in card.items() if any(s in k.lower() for s in ("power", "sclk", "mclk", "fclk", "temperature (sensor junction", "temperature (sensor memory", "performance"))}
except Exception as e:                      # noqa: BLE001
    out["smi"] = f"{type(e).__name__}: {e}"
print(json.dumps(out))
