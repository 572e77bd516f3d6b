#!/usr/bin/env python3
"""Package power and shader clock WHILE THE STEP RUNS ("power-limited" rested on a round-3 bare-MFMA stream).
Runs, ~4 s each and back to back with ~2 s of idle between: idle, the bare fp16 MFMA calibration stream (random operands), the
config-3 forward step (B = 4096), the stand-alone quantizer, the config-3 step on three-term bf16 -- sampling power / sclk from the
amdgpu hwmon files (or rocm-smi where they are missing) every 50 ms in a thread.    -> one table on stdout"""
import glob, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vqvae_amd import _lib, conv as conv_mod, conv_hip, functional as F
from vqvae_amd.modules import VQVAE

dev = torch.device("cuda:0")
S = bench.PowerSampler(bench.PowerSampler.pci_of(torch, dev))
print("device pci:", S.pci, torch.cuda.get_device_properties(dev).name)
print("sampler:", S.describe(), flush=True)
conv_mod.set_conv_backend("hip")
torch.manual_seed(0)
m = VQVAE(128, 32, 2, 512, 64, 0.25).eval().to(dev)
x = torch.randn(4096, 3, 32, 32, device=dev)
with torch.no_grad():
    z_e = conv_hip.encoder_forward(m.encoder, x, m.pre_quantization_conv)
    cbw = m.vector_quantization.embedding.weight.detach()
    vws = F.vq_workspace(512, 64, dev)
    F.vq_forward(z_e, cbw, 0.25, rowmajor=True, workspace=vws)
L = _lib.load()
nb = L.vqvae_calibration_scratch_bytes()
scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream


def loop(fn, seconds, per_call_sync_every=20):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(per_call_sync_every):
            fn()
        torch.cuda.synchronize()
        n += per_call_sync_every
    return n, time.perf_counter() - t0


def phase(name, fn, seconds=4.0, unit_ms=True):
    torch.cuda.synchronize()
    time.sleep(2.0)
    with torch.no_grad():
        loop(fn, 0.5)                                  # ramp
        S.start()
        n, el = loop(fn, seconds)
        r = S.stop()
    print(f"{name:44s} {el / n * 1e3:9.4f} ms/call   power avg {r['power_w_avg']} W (min {r['power_w_min']}, max {r['power_w_max']})   "
          f"sclk avg {r['sclk_mhz_avg']} MHz (min {r['sclk_mhz_min']}, max {r['sclk_mhz_max']})   {r['samples']} samples", flush=True)


time.sleep(2.0)
S.start(); time.sleep(3.0); r = S.stop()
print(f"{'idle':44s} {'':9s}           power avg {r['power_w_avg']} W   sclk avg {r['sclk_mhz_avg']} MHz   {r['samples']} samples")
phase("bare fp16 MFMA stream (calibration kernel)", lambda: L.vqvae_calibration_mfma_f16(2000, scratch.data_ptr(), nb, st))
phase("config-3 forward, two-term fp16 (headline)", lambda: m(x))
phase("stand-alone quantizer, 262 144 rows", lambda: F.vq_forward(z_e, cbw, 0.25, rowmajor=True, workspace=vws, prepared=True))
phase("config-3 forward, three-term bf16", lambda: m._forward_c(x, fwd_flags=F.FWD_CONV_BF16_SPLIT))
phase("config-3 forward, exact fp32 MFMA", lambda: m._forward_c(x, fwd_flags=F.FWD_CONV_EXACT_FP32))
phase("config-3 forward, two-term fp16 (again)", lambda: m(x))
print("rocm-smi limits:", end=" ")
try:
    j = json.loads(subprocess.run(["rocm-smi", "--showmaxpower", "--showperflevel", "--json"], capture_output=True, text=True, timeout=30).stdout)
    print(next(iter(j.values())))
except Exception as e:      # noqa: BLE001
    print(type(e).__name__, e)
