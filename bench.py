#!/usr/bin/env python3
"""bench.py -- images/sec of the VQ-VAE forward on N MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5] [--batch B]

One process per GPU.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes this file under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (so the plain command
produces an N-rank line); launched by a driver that already set RANK / WORLD_SIZE it runs as that rank, and
WORLD_SIZE != --gpus is an error.  The batch shards embarrassingly: every rank runs the same per-GPU batch on its own
replica of the weights (weak scaling), there is no data-path collective; RCCL carries only the barrier and the
max-over-ranks of the elapsed time.

A "step" is one `VQVAE.forward(x)` over one synthetic batch already resident in HBM:
  c3 (default)  BASELINE config 3: full HIP path (Encoder + VQ + Decoder kernels), 32x32x3,  K=512,  D=64,  B=4096/GPU
  c2            BASELINE config 2: HIP VectorQuantizer, torch (MIOpen) convs,       32x32x3,  K=512,  D=64,  B=1024/GPU
  c4            BASELINE config 4: full HIP path,                                   224x224x3, K=1024, D=64,  B=512/GPU
  c5            BASELINE config 5: full HIP path, per-GPU shard of B=8192 over 8,   256x256x3, K=8192, D=128, B=1024/GPU
There is no fallback between workloads: a missing kernel library or extension is an error.

Timing: W warm-up steps, then R repeats of EXACTLY K steps, each repeat bracketed by barrier + synchronize on both
sides and reduced with MAX over ranks; R is chosen so that the timed work is at least ~1 s (never fewer than 5
repeats).  `value` / `ms_per_step` come from the MEDIAN repeat; min / max are reported beside it.

Rank 0 prints ONE JSON line.  `roofline` is for the fused VectorQuantizer kernel, timed live with HIP events on the
launch stream (vqvae_profile_* hooks) in extra steps after the timed region -- the event pairs add a little
overhead, so the per-kernel figures sum to slightly more than one un-instrumented step.  On the default shapes the step's
quantizer runs inside the encoder's last kernel; `roofline` then times the standalone quantizer kernel on the same z_e.
`index_flips_vs_reference` = every index of this batch against the reference's algorithm on the host;
`other_workloads` = BASELINE configs 2 / 4 / 5 on this GPU, >= 1 s timed each; `training_step` = SURVEY 8(f) row 2
(main.py:74-78: forward + losses + backward on the HIP kernels, no optimizer) at the same batch.  `roofline.traffic` is
HBM bytes per launch from rocprofv3 PMC passes recorded in the file named by `traffic_source` (null when no
such file is committed for the workload); it is never a constant in this script.  `cpu_baseline` is the reference's
algorithm on the host cores (oracle/torch_port.py, same ATen ops as the reference, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_ACHIEVABLE_GBPS = 6290.0
MFMA_16BIT_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA
MFMA_FP32_PEAK_TFLOPS = 157.3     # fp32 MFMA (256 flop / clk / CU x 256 CUs x 2.4 GHz)
MFMA_F32_PEAK_TFLOPS = 157.3      # exact-fp32 MFMA = vector rate
CALIB_REFERENCE_TFLOPS = 1700.0   # what the calibration stream sustained on the box of profiles/r03_mfma_power.txt (1 701 TF at 1.69 GHz)

# name -> (description, per-GPU batch, H = W, K, D, conv backend)
WORKLOADS = {
    "c3": ("BASELINE config 3: full HIP path (Encoder+VQ+Decoder), 32x32x3, K=512, D=64", 4096, 32, 512, 64, "hip"),
    "c2": ("BASELINE config 2: HIP VectorQuantizer kernel, torch (MIOpen) convs unchanged, 32x32x3, K=512, D=64",
           1024, 32, 512, 64, "torch"),
    "c4": ("BASELINE config 4: full HIP path, 224x224x3 -> 56x56 latent, K=1024, D=64", 512, 224, 1024, 64, "hip"),
    "c5": ("BASELINE config 5: full HIP path, per-GPU shard (1024 images) of the 8192-image batch, 256x256x3, "
           "K=8192, D=128", 1024, 256, 8192, 128, "hip"),
}


def conv_flops_per_image(H, W, D, h_dim=128, res_h=32, n_res=2, ends=False):
    """2*MAC of every conv / conv-transpose / residual layer between the first and the last; ends=True:
    plus the first (4x4 s2, 3 -> h/2) and the last (conv-transpose 4x4 s2, h/2 -> 3) layer."""
    h2, h4 = (H // 2) * (W // 2), (H // 4) * (W // 4)
    if ends:
        first = 2 * h2 * 3 * 16 * (h_dim // 2)
        last = 2 * (H * W) * (h_dim // 2) * 4 * 3              # 4 phases x 2x2 taps per output pixel
        return conv_flops_per_image(H, W, D, h_dim, res_h, n_res) + first + last
    enc2 = 2 * h4 * (h_dim // 2) * 16 * h_dim                  # 4x4 s2, 64 -> 128
    enc4 = 2 * h4 * h_dim * 9 * h_dim                          # 3x3, 128 -> 128
    res = n_res * 2 * h4 * (h_dim * 9 * res_h + res_h * h_dim)  # per stack
    pre = 2 * h4 * h_dim * D
    dec0 = 2 * h4 * D * 9 * h_dim
    dec2 = 2 * h2 * h_dim * 4 * (h_dim // 2)                   # convT 4x4 s2 = 4 phases x 2x2 taps
    return enc2 + enc4 + 2 * res + pre + dec0 + dec2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="lower bound on the timed work (sets the repeats)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-power", action="store_true", help="skip the ~3.5 s power / clock sampling run behind the timed region")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the appended measurements of BASELINE configs 2 / 4 / 5 (default line, 1 GPU, c3 only)")
    ap.add_argument("--dry-run", action="store_true",
                    help="control-flow check without a GPU (gloo, a stub CPU step): exercises the rank spawn, the "
                         "barriers, the max-reduce and the JSON line; its numbers are NOT measurements")
    return ap.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn(args):
    """`python bench.py --gpus N` from a plain shell: become N ranks (one per GPU) under torch.distributed.run."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def cpu_baseline(seconds: float, torch):
    """Reference algorithm on the host: same ATen CPU ops in the same order as the reference
    (oracle/torch_port.py), main.py defaults (32x32x3, K=512, D=64), eval + no_grad, on a bounded sample.  A few
    thread counts are tried briefly (big hosts thrash on 32-image batches) and the best is reported with the
    thread count that produced it."""
    from oracle import torch_port
    sd = torch_port.init_state_dict()
    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (8, 16, 32, 64, torch.get_num_threads()) if t <= ncpu})
    per = max(0.5, seconds / (2 * len(cands)))
    best, best_t, detail = 0.0, cands[0], []
    for t in cands:
        torch.set_num_threads(t)
        for B in (32, 256):
            x = torch.randn(B, 3, 32, 32)
            torch_port.forward(sd, x, 0.25, 2)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < per:
                torch_port.forward(sd, x, 0.25, 2)
                n += 1
            ips = n * B / (time.perf_counter() - t0)
            detail.append(f"T={t},B={B}:{ips:.0f}")
            if ips > best:
                best, best_t = ips, t
    # SURVEY.md 8d's own protocol point beside the sweep: B = 32 (BASELINE config 1), 5 warm-ups + 50 timed
    # forwards, median -- with the port standing in for the unmodified reference, which cannot travel to the GPU box (bitwise equal
    # to it in the build container: tests/test_oracle.py).  Threads: the survey says nproc; on a host whose nproc counts SMT siblings
    # of a 2 x 64-core box (256) that is an oversubscription artefact (round 5 read 2.5 images/s from ONE sample), so the point runs on
    # the PHYSICAL cores and says so; fewer than 5 timed samples inside the wall-time bound are labelled as such.
    import statistics
    phys = ncpu
    try:
        cores_seen, pid, cid = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    cores_seen.add((pid, cid))
                pid = cid = None
        if cores_seen:
            phys = min(ncpu, len(cores_seen))
    except OSError:
        pass
    t8d = min(phys, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else phys)
    torch.set_num_threads(t8d)
    x = torch.randn(32, 3, 32, 32)
    nwarm, tw = 0, time.perf_counter()
    for _ in range(5):
        torch_port.forward(sd, x, 0.25, 2)
        nwarm += 1
        if time.perf_counter() - tw > 6.0:
            break
    ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        torch_port.forward(sd, x, 0.25, 2)
        ts.append(time.perf_counter() - t0)
        if sum(ts) > 12.0:
            break
    survey_8d = {"value": round(32 / statistics.median(ts), 1), "unit": "images/s", "batch": 32, "threads": t8d,
                 "threads_note": f"physical cores visible to this process ({ncpu} logical cpus on the host)",
                 "statistic": (f"median of {len(ts)} forwards after {nwarm} warm-ups" if len(ts) >= 5 else
                               f"ONLY {len(ts)} forwards fitted the 12 s bound: not a median, an oversubscribed host"),
                 "ms_per_iter": round(statistics.median(ts) * 1e3, 3)}
    torch.set_num_threads(best_t)
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(best, 1), "unit": "images/s", "cores": best_t, "host_logical_cpus": ncpu, "host_physical_cores": phys, "kind": "port",
            "kind_detail": "port (oracle/torch_port.py = the reference's ATen ops); `value` = BEST of a threads x batch sweep (generous to "
                           "the CPU); `cores` = the thread count of that best point; `survey_8d` = SURVEY 8d's own protocol point (B=32, one thread per physical "
                           "core, median of 50)",
            "survey_8d": survey_8d,
            "sample": "VQVAE.forward 32x32x3 K=512 D=64 fp32 eval/no_grad on the host CPU (" + cpu +
                      f", {ncpu} logical cpus); ~{per:.1f}s per (threads,batch) point, img/s: " + " ".join(detail)}


class PowerSampler:
    """Package power (W) and shader clock (MHz) of the GPU, sampled in a thread while something runs: the amdgpu hwmon files where the
    box exposes them (power1_average / power1_input in microwatt, freq1_input in Hz: a read costs microseconds), else `rocm-smi
    --showpower --showclocks --json` (~0.4 s per sample).  bench.py samples a SEPARATE sustained run of the step behind the timed
    region (a subprocess per sample beside the timed loop would be a perturbation of its own)."""

    def __init__(self, pci: str | None = None, period: float = 0.05):
        """pci: the device's PCI address ("0000:c1:00.0"; PowerSampler.pci_of(torch, dev)) -- a box shows the hwmon files of EVERY GPU of
        the node, also of those this process cannot open"""
        import glob
        self.period, self.samples, self._stop, self._thread = period, [], None, None
        self.power_file = self.freq_file = None
        self.pci = pci
        cards = sorted(glob.glob("/sys/class/drm/card*/device"))
        if pci:
            cards = [c for c in cards if os.path.basename(os.path.realpath(c)).lower() == pci.lower()]
        for hw in [h for c in cards for h in sorted(glob.glob(os.path.join(c, "hwmon/hwmon*")))]:
            for name in ("power1_average", "power1_input"):
                f = os.path.join(hw, name)
                if self.power_file is None and os.path.exists(f) and self._read(f) is not None:
                    self.power_file = f
            f = os.path.join(hw, "freq1_input")
            if self.freq_file is None and os.path.exists(f) and self._read(f) is not None:
                self.freq_file = f
            if self.power_file:
                break
        self.mode = "hwmon" if self.power_file else "rocm-smi"

    @staticmethod
    def pci_of(torch, dev):
        try:
            p = torch.cuda.get_device_properties(dev)
            return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except Exception:                          # noqa: BLE001
            return None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def describe(self):
        return (f"{self.mode} of {self.pci}: power={self.power_file} sclk={self.freq_file}" if self.mode == "hwmon" else
                "rocm-smi --showpower --showclocks --json")

    def _one(self):
        if self.mode == "hwmon":
            p = self._read(self.power_file)
            f = self._read(self.freq_file) if self.freq_file else None
            return (p / 1e6 if p is not None else None, f / 1e6 if f is not None else None)
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
            card = next(iter(json.loads(out).values()))
            p = next((float(v) for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), None)
            f = next((float(str(v).strip("()").lower().replace("mhz", "")) for k, v in card.items() if k.lower().startswith("sclk clock speed")), None)
            return (p, f)
        except Exception:                          # noqa: BLE001  (a sample is optional)
            return (None, None)

    def start(self):
        import threading
        self.samples, self._stop = [], threading.Event()

        def run():
            while not self._stop.is_set():
                self.samples.append(self._one())
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=15)
        ps = [p for p, _ in self.samples if p is not None]
        fs = [f for _, f in self.samples if f is not None]
        r1 = lambda v: round(v, 1)                  # noqa: E731
        return {"source": self.describe(), "samples": len(self.samples),
                "power_w_avg": r1(sum(ps) / len(ps)) if ps else None, "power_w_min": r1(min(ps)) if ps else None, "power_w_max": r1(max(ps)) if ps else None,
                "sclk_mhz_avg": r1(sum(fs) / len(fs)) if fs else None, "sclk_mhz_min": r1(min(fs)) if fs else None, "sclk_mhz_max": r1(max(fs)) if fs else None}


def step_power(step, torch, dev, seconds=3.0):
    """Average package power and shader clock while the step runs back to back for `seconds` (a run of its own behind the timed region)."""
    S = PowerSampler(PowerSampler.pci_of(torch, dev))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 0.5:          # ramp: the clock needs a moment of load to settle
        step()
    torch.cuda.synchronize()
    S.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        n += 20
    el = time.perf_counter() - t0
    r = S.stop()
    r["ms_per_step_during_sampling"] = round(el / n * 1e3, 4)
    r["seconds"] = round(el, 2)
    if r.get("sclk_mhz_avg"):
        # work per shader cycle: what stays of the step when the clock a box holds at its power cap is divided out
        r["us_per_step_x_ghz"] = round(el / n * 1e6 * r["sclk_mhz_avg"] / 1e3, 1)
    return r


def calibrate(torch, dev, seconds=0.6):
    """Box calibration: a bare fp16 MFMA stream on random operands (vqvae_calibration_mfma_f16, the kernel
    of tools/ubench/mfma_power.hip) for ~`seconds` right before the timed region.  The sustained rate of that stream is set by
    the clock the chip holds at its power limit, which differs from box to box by more than a round's kernel work does
    (profiles/r03_notes.txt section 10: 0.905 ... 1.09 ms per step for one build); with it in the line a reader can tell
    "box" from "build".  Returns TFLOP/s and the in-kernel shader clock of the LAST launch (the clock needs ~1 s of load to settle;
    the earlier launches bring the chip there)."""
    import ctypes
    from vqvae_amd import _lib
    L = _lib.load()
    nb = L.vqvae_calibration_scratch_bytes()
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    iters = 20000
    st = torch.cuda.current_stream(dev).cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, ms, launches = time.perf_counter(), 0.0, 0
    while True:
        e0.record()
        _lib.check(L.vqvae_calibration_mfma_f16(iters, scratch.data_ptr(), nb, st))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        launches += 1
        if time.perf_counter() - t0 >= seconds:
            break
    clk = scratch[:512 * 16].view(torch.int64).view(512, 2).double().cpu()
    ghz = float((clk[:, 0] / (clk[:, 1] * 10e-9)).mean() / 1e9)
    tf = L.vqvae_calibration_flops(iters) / (ms * 1e-3) / 1e12
    del scratch
    return {"mfma_fp16_random_tflops": round(tf, 1), "sclk_ghz": round(ghz, 3), "launches": launches,
            "seconds": round(time.perf_counter() - t0, 2), "reference_tflops": CALIB_REFERENCE_TFLOPS,
            "kernel": "vqvae_calibration_mfma_f16: bare v_mfma_f32_32x32x16_f16 stream, random operands, 2 waves/SIMD on every CU "
                      "(tools/ubench/mfma_power.hip); figures of the last launch"}


def source_sha():
    """Fingerprint of the kernel sources the running library was built from (vqvae_amd/csrc/*): ties a committed
    profile to the build it was taken on (vqvae_amd.build.source_fingerprint; `_lib.load()` refuses or rebuilds a library whose
    own fingerprint differs from it)."""
    from vqvae_amd import build as _build
    return _build.source_fingerprint()


def vq_instance_traffic(pmc, instance: str, rows: int):
    """HBM bytes per launch of ONE template instance of the quantizer at ONE row count, from the per-instance table
    tools/pmc_traffic.py writes (`vq_instances`: "<instance>@<rows>" -> read / write bytes per launch), or None: the figure in
    `roofline.traffic` must describe the kernel the line names and times, not another instance at another size."""
    if not pmc:
        return None
    import re

    def norm(name):                       # "vq_x<4, false, 1, 0>" (rocprofv3 prints defaulted arguments) == "vq_x<4, false, 1>"; drop remarks
        name = name.split(" (")[0]
        return re.sub(r"(, 0)+>$", ">", name)
    want = norm(instance)
    for key, ent in pmc.get("vq_instances", {}).items():
        inst, _, r = key.rpartition("@")
        if int(r) != rows:
            continue
        if norm(inst) == want:
            return int(ent["read_bytes"] + ent["write_bytes"])
    # kernels the library names without template arguments are the streamed-codebook FAMILY (four launches per slab, timed together):
    # all quantizer launches of one step
    ent = pmc.get("vq_instances", {}).get(f"vq_step@{rows}") if "<" not in want else None
    return int(ent["read_bytes"] + ent["write_bytes"]) if ent else None


def pmc_traffic(workload: str, vq_kernel: str):
    """HBM bytes per row / per image from the committed rocprofv3 PMC summary for this workload, or None.
    The file is written by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE doubled on
    gfx950, as MI355X_MICROARCH.md prescribes) and stamped with the fingerprint of the kernel sources it was measured on
    and the quantizer kernel's name: a file taken on other sources, or for another kernel, is NOT used (`traffic` null and
    `traffic_stale` says why) -- a traffic regression cannot hide behind an old file."""
    path = os.path.join(ROOT, "profiles", f"hbm_traffic_{workload}.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, None
    rel = os.path.relpath(path, ROOT)
    if d.get("source_sha") != source_sha():
        return None, f"{rel}: measured on kernel sources {d.get('source_sha')}, running {source_sha()}"
    # (files with a per-instance table are looked up by template instance and row count, vq_instance_traffic; older files carried one
    # kernel's figure and are only good for that kernel)
    if "vq_instances" not in d and vq_kernel and vq_kernel not in d.get("vq_kernel", ""):
        return None, f"{rel}: measured on {d.get('vq_kernel')!r}, running {vq_kernel}"
    d["_path"] = rel
    return d, None


def index_flips(model, x, torch, fwd_flags=0):
    """min_encoding_indices of the HIP path against the reference's algorithm (oracle/torch_port.py) on THIS batch: every
    row compared, the flips counted (z_e differs by conv rounding noise, so rows whose two best codes are closer than that
    may flip: SURVEY.md 8c expects <= 1e-4).  Checker only; runs after the timed region."""
    from oracle import torch_port
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        out = model._forward_c(x, want_idx=True, fwd_flags=fwd_flags)
        got = out[3].view(-1).cpu()
        xc = x.cpu()
        want = []
        for i in range(0, xc.shape[0], 512):
            z_e = torch_port.encode(sd, xc[i:i + 512].clone(), 2)
            want.append(torch_port.quantize(z_e, sd["vector_quantization.embedding.weight"], 0.25)[4].view(-1))
        want = torch.cat(want)
    n = int((got != want).sum())
    return {"flips": n, "rows": int(got.numel()), "rate": n / got.numel(), "expected_at_most": 1e-4,
            "reference": "oracle/torch_port.py (bitwise the imported reference) on the same batch, host CPU"}


def training_step(torch, dev, seconds=1.0):
    """SURVEY.md 8(f) row 2, the reference's main.py:74-78 without the optimizer: forward (saving activations), the fused
    losses, backward -- every conv, quantizer and gradient on libvqvae_hip.so.  Plus the dominant backward kernel alone:
    the weight gradient of the 3x3 128 -> 128 layer (exact fp32 products on the fp32 matrix cores) against that pipe's peak."""
    import statistics
    from vqvae_amd import autograd_conv, conv as conv_mod, training as T
    from vqvae_amd.modules import VQVAE
    desc, B, HW, K, D, _ = WORKLOADS["c3"]
    conv_mod.set_conv_backend("hip")
    torch.manual_seed(0)
    model = VQVAE(128, 32, 2, K, D, 0.25).to(dev).train()
    x = torch.randn(B, 3, HW, HW, generator=torch.Generator().manual_seed(1000)).to(dev)

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model.zero_grad(set_to_none=True)
            el, xh, pp = model(x)
            T.step_losses(el, xh, pp, x, 0.06)[1].backward()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(3)
    steps = max(2, int(seconds / 5 / max(run(2) / 2, 1e-6)) + 1)
    times = [run(steps) for _ in range(5)]
    el = statistics.median(times)
    res = {"workload": "main.py:74-78 (forward + recon/embedding losses + backward, no optimizer), 32x32x3, K=512, D=64",
           "per_gpu_batch": B, "images_per_s": round(B * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 4),
           "timed_seconds": round(sum(times), 3), "steps_per_repeat": steps}
    model.zero_grad(set_to_none=True)
    del model, x
    a = torch.randn(B, 8, 8, 128, device=dev)
    bt = torch.randn(B, 8, 8, 128, device=dev)
    for _ in range(2):
        autograd_conv.conv_wgrad(a, bt, 3, 1, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()                                        # the kernels run on torch's current stream (conv_hip._sp)
    for _ in range(n):
        autograd_conv.conv_wgrad(a, bt, 3, 1, 1)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e-3
    flops = 2.0 * B * 64 * 9 * 128 * 128
    # two-term fp16 products (round 4): three fp16 MFMA term products per fp32 product, priced against the dense fp16 peak
    res["weight_gradient_3x3_128"] = {"kernels": "conv_wgrad_map8_h2_kernel<3, 1, 2, 2> + conv_wgrad_reduce_kernel", "us": round(t * 1e6, 1),
                                      "bound": "mfma", "dtype": "f32 (products from two fp16 terms per operand, <= 2^-21 rel per product)",
                                      "achieved": round(3 * flops / t / 1e12, 1), "peak": MFMA_16BIT_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": round(3 * flops / t / 1e12 / MFMA_16BIT_PEAK_TFLOPS, 4),
                                      "term_products_per_mac": 3, "achieved_algorithmic_tflops": round(flops / t / 1e12, 1)}
    autograd_conv.WGRAD_EXACT_FP32 = True              # the same call on the exact-fp32 MFMA kernel (round 3's arithmetic), for the record
    try:
        autograd_conv.conv_wgrad(a, bt, 3, 1, 1)
        e0.record()
        for _ in range(n):
            autograd_conv.conv_wgrad(a, bt, 3, 1, 1)
        e1.record()
        torch.cuda.synchronize()
        t32 = e0.elapsed_time(e1) / n * 1e-3
        res["weight_gradient_3x3_128"]["exact_fp32_us"] = round(t32 * 1e6, 1)
        res["weight_gradient_3x3_128"]["exact_fp32_frac_of_fp32_mfma_peak"] = round(flops / t32 / 1e12 / MFMA_FP32_PEAK_TFLOPS, 4)
    finally:
        autograd_conv.WGRAD_EXACT_FP32 = False
    del a, bt
    torch.cuda.empty_cache()
    return res


def other_workload(name, torch, dev, seconds=1.0):
    """One of BASELINE's other single-GPU configurations, timed like the main line (>= `seconds` of timed steps, median
    of 5 repeats), with the live per-kernel figures of its quantizer and conv kernels."""
    import statistics
    from vqvae_amd import _lib, conv as conv_mod
    from vqvae_amd.modules import VQVAE
    desc, B, HW, K, D, conv_backend = WORKLOADS[name]
    conv_mod.set_conv_backend(conv_backend)
    torch.manual_seed(0)
    model = VQVAE(128, 32, 2, K, D, 0.25).eval().to(dev)
    x = torch.randn(B, 3, HW, HW, generator=torch.Generator().manual_seed(1000)).to(dev)

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(n):
                out = model(x)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    run(2)
    t1, _ = run(2)
    steps = max(2, int(seconds / 5 / max(t1 / 2, 1e-6)) + 1)
    times = [run(steps)[0] for _ in range(5)]
    el = statistics.median(times)
    _lib.profile_enable(True)
    nprof = min(steps, 10)
    run(nprof)
    vq_ms, vq_n = _lib.profile_collect("vq_main")
    conv_ms = 0.0
    for k in ("conv_igemm", "res_layer", "conv_in", "conv_out"):
        ms, n = _lib.profile_collect(k)
        conv_ms += ms
    _lib.profile_enable(False)
    rows = B * (HW // 4) * (HW // 4)
    res = {"workload": desc, "per_gpu_batch": B, "images_per_s": round(B * steps / el, 1),
           "ms_per_step": round(el / steps * 1e3, 4), "timed_seconds": round(sum(times), 3), "steps_per_repeat": steps}
    if vq_n:
        t_vq = vq_ms / vq_n * 1e-3
        # (VQVAE.forward hands the quantizer row-major rows with BOTH conv backends -- the torch backend permutes z_e itself,
        # vqvae_amd/conv.py -- so the kernel is the row-major one; round 3 printed the NCHW kernel's name here by mistake)
        res["vq"] = {"kernel": _lib.vq_kernel_instance(rows, K, D, (HW // 4) ** 2, 0x1), "avg_kernel_us": round(t_vq * 1e6, 2),
                     "hbm_GBps": round(rows * (8 * D + 8) / t_vq / 1e9, 1),
                     "hbm_frac": round(rows * (8 * D + 8) / t_vq / 1e9 / HBM_PEAK_GBPS, 4),
                     # at K >= 1024 the screen is matrix-bound, not HBM-bound
                     "screen_tflops_16bit": round(2.0 * rows * K * D / t_vq / 1e12, 1),
                     "mfma_frac": round(2.0 * rows * K * D / t_vq / 1e12 / MFMA_16BIT_PEAK_TFLOPS, 4)}
    if name == "c2" and vq_n:
        # The same launch STAND-ALONE on the step's own z_e (back-to-back launches, warm L2 / clocks) beside the in-situ figure above
        # (one launch per step between MIOpen's convs: the codebook image and the kernel's code are cold, the clock follows the convs'
        # load): VERDICT r5 weak 6 -- 41.5 us in situ against 17.5 us stand-alone on the driver's box.  README quotes the in-situ one
        # for config 2 (it is what the step pays) and says so.
        try:
            from vqvae_amd import conv_hip as _ch, functional as F_
            with torch.no_grad():
                z_rows = model.pre_quantization_conv(model.encoder(x)).permute(0, 2, 3, 1).contiguous() if conv_backend == "torch" else \
                    _ch.encoder_forward(model.encoder, x, model.pre_quantization_conv)
                cbw = model.vector_quantization.embedding.weight.detach().contiguous()
                ws_s = F_.vq_workspace(K, D, dev)
                F_.vq_forward(z_rows, cbw, 0.25, rowmajor=True, workspace=ws_s)
                for _ in range(3):
                    F_.vq_forward(z_rows, cbw, 0.25, rowmajor=True, workspace=ws_s, prepared=True)
                torch.cuda.synchronize()
                _lib.profile_enable(True)
                for _ in range(30):
                    F_.vq_forward(z_rows, cbw, 0.25, rowmajor=True, workspace=ws_s, prepared=True)
                ms_s, n_s = _lib.profile_collect("vq_main")
                _lib.profile_enable(False)
            t_s = ms_s / max(n_s, 1) * 1e-3
            res["vq"]["timing"] = "in situ: one launch per step between the torch (MIOpen) convs"
            res["vq_standalone"] = {"kernel": res["vq"]["kernel"], "avg_kernel_us": round(t_s * 1e6, 2),
                                    "hbm_frac": round(rows * (8 * D + 8) / t_s / 1e9 / HBM_PEAK_GBPS, 4),
                                    "timing": "30 back-to-back launches on the step's own z_e (warm L2, steady clock)",
                                    "in_situ_over_standalone": round(t_vq / t_s, 2)}
            del z_rows, ws_s
        except Exception as e:                               # noqa: BLE001
            _lib.profile_enable(False)
            res["vq_standalone"] = {"error": repr(e)[:200]}
    if name == "c2":
        # the same quantizer at the REFERENCE's own boundary: VectorQuantizer.forward(z) takes NCHW (models/quantizer.py:45-46, :74 are its
        # permute + copy passes) -- what integration/vqvae_hip_stub.py binds; timed by the dispatch's events like the row-major figure
        try:
            from vqvae_amd import functional as F_
            with torch.no_grad():
                z_nchw = torch.randn(B, D, HW // 4, HW // 4, generator=torch.Generator().manual_seed(7)).to(dev) * 0.07
                cbw = model.vector_quantization.embedding.weight.detach().contiguous()
                ws_n = F_.vq_workspace(K, D, dev)
                F_.vq_forward(z_nchw, cbw, 0.25, rowmajor=False, workspace=ws_n)
                torch.cuda.synchronize()
                _lib.profile_enable(True)
                for _ in range(20):
                    F_.vq_forward(z_nchw, cbw, 0.25, rowmajor=False, workspace=ws_n, prepared=True)
                ms_n, n_n = _lib.profile_collect("vq_main")
                _lib.profile_enable(False)
            if n_n:
                t_n = ms_n / n_n * 1e-3
                res["vq_nchw_boundary"] = {"kernel": _lib.vq_kernel_instance(rows, K, D, (HW // 4) ** 2, 0x0),
                                           "avg_kernel_us": round(t_n * 1e6, 2),
                                           "hbm_frac": round(rows * (8 * D + 8) / t_n / 1e9 / HBM_PEAK_GBPS, 4),
                                           "data": "N(0, 0.07^2) z_e in the module's (B, D, H, W) layout, this model's codebook"}
        except Exception as e:                               # (a reporting extra: never take the bench line down with it)
            _lib.profile_enable(False)
            res["vq_nchw_boundary"] = {"error": repr(e)[:200]}
    if conv_backend == "hip" and conv_ms > 0:
        t_conv = conv_ms / nprof * 1e-3
        flops_img = conv_flops_per_image(HW, HW, D, ends=True)
        terms = _lib.load().vqvae_conv_term_products(1, HW // 4, HW // 4, 128, 128, 0x100) or 6
        alg_tf = B * flops_img / t_conv / 1e12
        res["conv"] = {"ms_per_step": round(t_conv * 1e3, 4), "term_products_per_mac": terms,
                       "achieved_algorithmic_tflops": round(alg_tf, 1), "issued_tflops_16bit": round(terms * alg_tf, 1),
                       "mfma_frac": round(terms * alg_tf / MFMA_16BIT_PEAK_TFLOPS, 4), "timing": "hip-event brackets (event_inflated)"}
    del model, x
    torch.cuda.empty_cache()
    return res


# What the quantizer's launch shape can reach when it does NOTHING but move its 520 B per row (one workgroup per CU, codebook image
# into LDS, rows in, rows + indices out): tools/ubench/vq_floor.hip, profiles/r04b_vq_timeline.txt section 1, as a fraction of the
# 8 TB/s the roofline is priced against.  The distance between `frac` and this is the kernel's own; the rest is the machine's.
VQ_COPY_ROOF = {65536: 0.63, 262144: 0.97, 2097152: 0.66}


def vq_size_sweep(torch, dev, K=512, D=64):
    """The stand-alone quantizer at 65 536 / 262 144 / 2 097 152 rows of this model's own z_e (repeated to size), each timed by the
    dispatch's own events: fraction of the 8 TB/s peak beside the bare-copy roof of the same launch shape at that size."""
    from vqvae_amd import _lib, conv as conv_mod, conv_hip, functional as F_hip
    from vqvae_amd.modules import VQVAE
    conv_mod.set_conv_backend("hip")
    torch.manual_seed(0)
    model = VQVAE(128, 32, 2, K, D, 0.25).eval().to(dev)
    out = {}
    with torch.no_grad():
        x = torch.randn(1024, 3, 32, 32, generator=torch.Generator().manual_seed(1000)).to(dev)
        z1 = conv_hip.encoder_forward(model.encoder, x, model.pre_quantization_conv).reshape(-1, D)       # 65 536 rows
        cbw = model.vector_quantization.embedding.weight.detach()
        vws = F_hip.vq_workspace(K, D, dev)
        for mult in (1, 4, 32):
            z = z1.repeat(mult, 1).view(mult * 1024, 8, 8, D).contiguous()
            rows = z.shape[0] * 64
            F_hip.vq_forward(z, cbw, 0.25, rowmajor=True, workspace=vws)
            for _ in range(3):
                F_hip.vq_forward(z, cbw, 0.25, rowmajor=True, workspace=vws, prepared=True)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            n = 30 if mult < 32 else 12
            for _ in range(n):
                F_hip.vq_forward(z, cbw, 0.25, rowmajor=True, workspace=vws, prepared=True)
            ms, cnt = _lib.profile_collect("vq_main")
            _lib.profile_enable(False)
            t = ms / max(cnt, 1) * 1e-3
            frac = rows * (8 * D + 8) / t / 1e9 / HBM_PEAK_GBPS
            out[str(rows)] = {"kernel": _lib.vq_kernel_instance(rows, K, D, 64, 0x1), "avg_kernel_us": round(t * 1e6, 2),
                              "frac": round(frac, 4), "copy_roof_frac": VQ_COPY_ROOF.get(rows),
                              "frac_of_copy_roof": round(frac / VQ_COPY_ROOF[rows], 4) if rows in VQ_COPY_ROOF else None}
            del z
    del model, x, z1, vws
    torch.cuda.empty_cache()
    return out


def wire_legs(torch, dev, fwd_ms, seconds=0.6, B=4096, K=512, D=64):
    """SURVEY.md 8f-1, the index wire format as single entry points at the headline's batch: x -> indices (vqvae_encode_f32: the
    encoder's last kernel writes 512 B of indices per image, no z_e, no z_q) and indices -> x_hat (vqvae_decode_f32: the decoder's
    first kernel gathers the codebook rows itself)."""
    import statistics
    from vqvae_amd import conv as conv_mod
    from vqvae_amd.modules import VQVAE
    conv_mod.set_conv_backend("hip")
    torch.manual_seed(0)
    model = VQVAE(128, 32, 2, K, D, 0.25).eval().to(dev)
    x = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(1000)).to(dev)
    res = {}
    with torch.no_grad():
        idx = model.encode(x)
        # decode: the entry point itself (validate=False: indices that come from encode() need no host-side range check) and the
        # module's default call, which checks idx.min() / idx.max() on the host first as the reference's scatter would raise
        for name, fn in (("encode", lambda: model.encode(x)), ("decode", lambda: model.decode_indices(idx, B, 8, 8, validate=False)),
                         ("decode_validated", lambda: model.decode_indices(idx, B, 8, 8))):
            def run(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            run(3)
            steps = max(3, int(seconds / 5 / max(run(3) / 3, 1e-6)) + 1)
            el = statistics.median([run(steps) for _ in range(5)])
            ms = el / steps * 1e3
            res[name] = {"images_per_s": round(B * steps / el, 1), "ms_per_call": round(ms, 4),
                         "vs_forward": round(fwd_ms / ms, 2) if fwd_ms else None}
    res["encode"]["boundary_bytes_per_image"] = {"in": 3 * 32 * 32 * 4, "out": 64 * 8}
    res["decode"]["boundary_bytes_per_image"] = {"in": 64 * 8, "out": 3 * 32 * 32 * 4}
    res["decode_validated"]["boundary_bytes_per_image"] = res["decode"]["boundary_bytes_per_image"]
    res["note"] = ("decode = vqvae_decode_f32 alone (no host sync; a bad index would show as NaN pixels); decode_validated adds the Python "
                   "layer's range check (one min / max reduction and a host sync per call: the reference's scatter raises on a bad "
                   "index, a launch cannot)")
    del model, x, idx
    torch.cuda.empty_cache()
    return res


HPARAM_LEGS = {
    # main.py:16-25's other hyper-parameters on 32x32 images: (h_dim, res_h, n_res, K, D, batch, what runs)
    "wide_h256_rh64": (256, 64, 2, 512, 64, 1024,
                       "main.py --n_hiddens 256 --n_residual_hiddens 64 (h_dim 256, res_h 64, 2 residual layers), 32x32x3, K=512, D=64: "
                       "per-layer kernels, residual layers outside the fused widths as conv -> conv -> combine"),
    "k256": (128, 32, 2, 256, 64, 4096,
             "main.py --n_embeddings 256, 32x32x3, D=64: the four fused conv kernels, quantizer inside the encoder's last kernel "
             "(32 KiB codebook image)"),
    "k1024": (128, 32, 2, 1024, 64, 4096,
              "main.py --n_embeddings 1024, 32x32x3, D=64: the four fused conv kernels, quantizer inside the encoder's last kernel "
              "(128 KiB codebook image streamed through the weight stages in eight parts; before the second session of round 4 "
              "z_e was written and the streamed-codebook kernels quantized it)"),
    "d48": (128, 32, 2, 512, 48, 4096,
            "main.py --embedding_dim 48, 32x32x3, K=512 (round 5: any embedding width up to 256): conv kernels outside the 64-channel fused "
            "forms, the quantizer on the exact-fp32 vector kernel vq_generic_kernel (correct, not fast)"),
}


def hparam_leg(name, torch, dev, seconds=1.0):
    """One of HPARAM_LEGS through the module mirror's forward (one vqvae_forward_f32 call per step)."""
    import statistics
    from vqvae_amd import _lib, conv as conv_mod
    from vqvae_amd.modules import VQVAE
    h, rh, nres, K, D, B, desc = HPARAM_LEGS[name]
    conv_mod.set_conv_backend("hip")
    torch.manual_seed(0)
    model = VQVAE(h, rh, nres, K, D, 0.25).eval().to(dev)
    x = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(1000)).to(dev)

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(n):
                model(x)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(2)
    steps = max(2, int(seconds / 5 / max(run(2) / 2, 1e-6)) + 1)
    times = [run(steps) for _ in range(5)]
    el = statistics.median(times)
    res = {"workload": desc, "standalone_vq_kernel_for_K": _lib.vq_kernel_name(K, D, 0x1),
           "per_gpu_batch": B, "images_per_s": round(B * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 4),
           "timed_seconds": round(sum(times), 3), "steps_per_repeat": steps}
    del model, x
    torch.cuda.empty_cache()
    return res


def scheme_leg(scheme, torch, dev, seconds=1.0):
    """BASELINE config 3 with the whole path on another product scheme (the headline beside the same step in
    exacter arithmetic, same box, same batch): 'bf16x3' = VQVAE_FWD_CONV_BF16_SPLIT (three-term bf16 products, per-layer kernels),
    'fp32' = VQVAE_FWD_CONV_EXACT_FP32 (exact-fp32 MFMA kernels: the reference's arithmetic up to summation order)."""
    import statistics
    from vqvae_amd import conv as conv_mod, functional as F_hip
    from vqvae_amd.modules import VQVAE
    desc, B, HW, K, D, _ = WORKLOADS["c3"]
    flags = {"bf16x3": F_hip.FWD_CONV_BF16_SPLIT, "fp32": F_hip.FWD_CONV_EXACT_FP32}[scheme]
    conv_mod.set_conv_backend("hip")
    torch.manual_seed(0)
    model = VQVAE(128, 32, 2, K, D, 0.25).eval().to(dev)
    x = torch.randn(B, 3, HW, HW, generator=torch.Generator().manual_seed(1000)).to(dev)

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(n):
                model._forward_c(x, fwd_flags=flags)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(2)
    steps = max(2, int(seconds / 5 / max(run(2) / 2, 1e-6)) + 1)
    times = [run(steps) for _ in range(5)]
    el = statistics.median(times)
    res = {"workload": desc + {"bf16x3": " -- every conv layer on three-term bf16 products (6 MFMAs per fp32 product, <= 3 * 2^-24 rel "
                                         "per product, no operand scales), per-layer kernels",
                               "fp32": " -- every conv layer on the exact-fp32 MFMA kernels (v_mfma_f32_32x32x2_f32)"}[scheme],
           "flag": {"bf16x3": "VQVAE_FWD_CONV_BF16_SPLIT", "fp32": "VQVAE_FWD_CONV_EXACT_FP32"}[scheme],
           "per_gpu_batch": B, "images_per_s": round(B * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 4),
           "timed_seconds": round(sum(times), 3), "steps_per_repeat": steps,
           "index_flips_vs_reference": index_flips(model, x, torch, flags)}
    del model, x
    torch.cuda.empty_cache()
    return res


def trained_leg(name, torch, dev, seconds=1.0):
    """BASELINE config 3 on a REALLY TRAINED checkpoint (round 6; tests/golden/<name>_state.npz: main.py:67-98's loop on the HIP
    training path, tools/train_checkpoint.py) and structured images (tests/synthdata.py), through the module's DEFAULT call: the
    product scheme is whatever vqvae_weights_range_check_f32 recommends for this checkpoint.  Beside the rate: the guard's per-layer
    spreads and its pick, every index against the reference's algorithm, the stand-alone quantizer's time on THIS z_e (its speed is
    data-dependent: open / hard rows take the exact part), and the row classes of that z_e measured with the VQ_DEBUG_VERDICT build
    (tools/r06_row_classes.py -> profiles/vq_row_classes.json; not measurable in the product library)."""
    import statistics
    from tests import cases, synthdata
    from vqvae_amd import _lib, conv as conv_mod, conv_hip, functional as F_hip
    from vqvae_amd.modules import VQVAE
    desc, B, HW, K, D, _ = WORKLOADS["c3"]
    h, rh, nl, K, D, beta, _, seed = cases.TRAINED_CASES[name]
    conv_mod.set_conv_backend("hip")
    model = VQVAE(h, rh, nl, K, D, beta).eval()
    model.load_state_dict(cases.trained_state(name))
    model = model.to(dev)
    x = synthdata.normalised(B, seed + 7).to(dev)
    flags, spreads = model.scheme_hint()

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(n):
                model(x)                                   # the default call: the guard's scheme
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(2)
    steps = max(2, int(seconds / 5 / max(run(2) / 2, 1e-6)) + 1)
    times = [run(steps) for _ in range(5)]
    el = statistics.median(times)
    with torch.no_grad():
        loss, _, ppl, idx = model._forward_c(x, want_idx=True)
        z_e = conv_hip.encoder_forward(model.encoder, x, model.pre_quantization_conv)
        cbw = model.vector_quantization.embedding.weight.detach()
        vws = F_hip.vq_workspace(K, D, dev)
        F_hip.vq_forward(z_e, cbw, beta, rowmajor=True, workspace=vws)
        for _ in range(3):
            F_hip.vq_forward(z_e, cbw, beta, rowmajor=True, workspace=vws, prepared=True)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(30):
            F_hip.vq_forward(z_e, cbw, beta, rowmajor=True, workspace=vws, prepared=True)
        ms, cnt = _lib.profile_collect("vq_main")
        _lib.profile_enable(False)
    rows = B * 64
    t_vq = ms / max(cnt, 1) * 1e-3
    classes = None
    try:
        with open(os.path.join(ROOT, "profiles", "vq_row_classes.json")) as f:
            classes = json.load(f).get(name)
    except (OSError, ValueError):
        pass
    res = {"workload": desc + f" -- weights: {name} (trained, tests/golden/{name}_state.npz), images: tests/synthdata.py",
           "scheme_picked_by_guard": "bf16x3 (VQVAE_FWD_CONV_BF16_SPLIT)" if flags else "fp16x2 (default two-term fp16 products)",
           "guard_input_channel_spread_binades": [round(v, 2) for v in spreads], "guard_limit_binades": 10.0,
           "per_gpu_batch": B, "images_per_s": round(B * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 4),
           "timed_seconds": round(sum(times), 3), "steps_per_repeat": steps,
           "embedding_loss": round(float(loss), 6), "perplexity": round(float(ppl), 3), "codes_in_use": int(idx.unique().numel()),
           "index_flips_vs_reference": index_flips(model, x, torch, flags),
           "standalone_vq_on_this_z_e": {"kernel": _lib.vq_kernel_instance(rows, K, D, 64, 0x1), "avg_kernel_us": round(t_vq * 1e6, 2),
                                         "frac": round(rows * (8 * D + 8) / t_vq / 1e9 / HBM_PEAK_GBPS, 4)},
           "row_classes": classes if classes else "profiles/vq_row_classes.json has no entry for this checkpoint"}
    del model, x, z_e, vws
    torch.cuda.empty_cache()
    return res


def pixelcnn_leg(torch, dev, seconds=0.6):
    """one GatedPixelCNN forward (pixelcnn/models.py:118-127; 15 gated layers, dim 64, 512 codes) over the
    8x8 latent index maps of 1024 images, and the ancestral sampler (:129-146, 64 forwards) for 64 samples replayed from a hipGraph."""
    import statistics
    from vqvae_amd.pixelcnn import GatedPixelCNN
    torch.manual_seed(0)
    m = GatedPixelCNN(512, 64, 15, 10).to(dev).eval()
    B = 1024
    x = torch.randint(0, 512, (B, 8, 8), device=dev)
    label = torch.randint(0, 10, (B,), device=dev)

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(n):
                m(x, label)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(2)
    steps = max(2, int(seconds / 5 / max(run(2) / 2, 1e-6)) + 1)
    el = statistics.median([run(steps) for _ in range(5)])
    lab64 = torch.arange(10, device=dev).repeat(7)[:64]
    with torch.no_grad():
        m.generate(lab64, (8, 8), 64, use_graph=True)          # capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.generate(lab64, (8, 8), 64, use_graph=True)
        torch.cuda.synchronize()
        gen = time.perf_counter() - t0
    res = {"workload": "GatedPixelCNN(512 codes, dim 64, 15 layers, 10 classes) forward over 8x8 index maps",
           "per_gpu_batch": B, "ms_per_forward": round(el / steps * 1e3, 3), "maps_per_s": round(B * steps / el, 1),
           "generate_64_samples_8x8_ms": round(gen * 1e3, 1), "steps_per_repeat": steps}
    del m, x
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(respawn(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         "(or drop the launcher and let --gpus spawn them)")

    # one slice of the host's cores per rank (LOCAL_RANK of LOCAL_WORLD_SIZE): eight ranks' launch threads and their helper threads do
    # not migrate over each other; a no-op for a single rank or where the kernel forbids it
    try:
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if lws > 1 and hasattr(os, "sched_setaffinity"):
            cpus = sorted(os.sched_getaffinity(0))
            per = len(cpus) // lws
            if per >= 1:
                os.sched_setaffinity(0, cpus[local * per:(local + 1) * per])
    except OSError:
        pass
    import torch
    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the HIP path has no CPU fallback (use --dry-run for the control-flow check)")
    # VQVAE_BENCH_SHARE_GPU=1: all ranks on device 0 (N>1 control flow on a 1-GPU box); never used by the driver
    share_gpu = os.environ.get("VQVAE_BENCH_SHARE_GPU") == "1"
    backend = "gloo" if dry else os.environ.get("VQVAE_BENCH_DIST_BACKEND", "nccl")
    if share_gpu:
        local = 0
    dev = torch.device("cpu")
    if not dry:
        if local >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    # VQVAE_BENCH_FORCE_DIST=1: initialise the process group even for ONE rank -- RCCL refuses two ranks on one device
    # ("Duplicate GPU detected", profiles/r03_rccl_2rank_shared_gpu.log), so on a 1-GPU box this is how the NCCL branch
    # (init with device_id, barrier, device-side MAX all-reduce) gets executed at all
    if world > 1 or os.environ.get("VQVAE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist_mod.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_mod

    def sync():
        if not dry:
            torch.cuda.synchronize()

    desc, B0, HW, K, D, conv_backend = WORKLOADS[args.workload]
    B = args.batch or B0
    H = W = HW
    if dry:
        B, H, W = 8, 32, 32
        conv = torch.nn.Conv2d(3, 8, 3, padding=1)

        def step():                                   # stub: control flow only
            with torch.no_grad():
                return None, conv(x), None
        _lib = None
    else:
        from vqvae_amd import _lib, conv as conv_mod
        from vqvae_amd.modules import VQVAE
        _lib.load()                                    # fail loudly if the HIP library is missing
        if conv_backend == "hip":
            from vqvae_amd import conv_hip  # noqa: F401   (an ImportError here is an error, not a smaller workload)
        conv_mod.set_conv_backend(conv_backend)
        torch.manual_seed(0)                           # same weights on every rank (replicas)
        model = VQVAE(128, 32, 2, K, D, 0.25).eval().to(dev)

        def step():
            with torch.no_grad():
                return model(x)
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(B, 3, H, W, generator=g).to(dev)   # synthetic, resident in HBM before timing

    own_times = []

    def timed_repeat():
        sync()
        if dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        sync()
        if dist:
            dist.barrier()
        sync()
        el = time.perf_counter() - t0
        own_times.append(el)                            # this rank's own clock (the line lists every rank's median: a straggler shows)
        if dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    calib = None
    if not dry:
        calib = calibrate(torch, dev)                  # every rank: the GPUs enter the timed region equally warm
    for _ in range(args.warmup):
        step()
    first, out = timed_repeat()
    # every rank must run the same number of repeats: derive it from the max-reduced first repeat
    repeats = max(5, min(200, int(args.min_seconds / max(first, 1e-6)) + 1))
    if dry:
        repeats = 5
    times = [first]
    for _ in range(repeats - 1):
        el, out = timed_repeat()
        times.append(el)
    assert torch.isfinite(out[1]).all()
    srt = sorted(times)
    elapsed = srt[len(srt) // 2]                       # median repeat
    # every rank's own median step time and calibration, gathered on all ranks (tiny): rank 0 prints them
    own_ms = sorted(own_times)[len(own_times) // 2] / args.steps * 1e3
    per_rank = [{"rank": rank, "ms_per_step": round(own_ms, 4), "calibration_tflops": (calib or {}).get("mfma_fp16_random_tflops"),
                 "cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}]
    if dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered

    # ---- per-kernel time of the fused VQ kernel (and the conv groups), extra steps outside the timed region ----
    extra, vq_ms, vq_n, nprof = {}, 0.0, 0, min(args.steps, 50)
    if not dry:
        _lib.profile_enable(True)
        for _ in range(nprof):
            step()
        vq_ms, vq_n = _lib.profile_collect("vq_main")
        for name in ("conv_igemm", "res_layer", "conv_in", "conv_out"):
            ms, n = _lib.profile_collect(name)
            if n:
                extra[name] = {"ms_per_step": round(ms / nprof, 4), "launches_per_step": n // nprof}
        vq_in_step = vq_n > 0
        if not vq_in_step and conv_backend == "hip":
            # The step's quantizer runs INSIDE the encoder's last kernel (32x32 images, K = 512, D = 64: z_e never leaves the
            # chip), so it has no launch of its own to time.  The north-star grades the standalone fused-VQ kernel: time the
            # kernel vqvae_vq_forward_f32 launches, on this batch's own z_e (encoder entry point without the quantizer).
            from vqvae_amd import conv_hip, functional as F_hip
            with torch.no_grad():
                z_e = conv_hip.encoder_forward(model.encoder, x, model.pre_quantization_conv)        # (B, h, w, D) row-major
                cbw = model.vector_quantization.embedding.weight.detach()
                vws = F_hip.vq_workspace(K, D, dev)
                F_hip.vq_forward(z_e, cbw, 0.25, rowmajor=True, workspace=vws)
                for _ in range(3):
                    F_hip.vq_forward(z_e, cbw, 0.25, rowmajor=True, workspace=vws, prepared=True)
                _lib.profile_collect("vq_main")
                for _ in range(nprof):
                    F_hip.vq_forward(z_e, cbw, 0.25, rowmajor=True, workspace=vws, prepared=True)
                vq_ms, vq_n = _lib.profile_collect("vq_main")
            del z_e, vws
        _lib.profile_enable(False)

    if rank == 0:
        n_gpus = world
        line = {
            "metric": "images/sec VQ-VAE forward (32x32x3, K=512, D=64)" if args.workload in ("c2", "c3") else
                      f"images/sec VQ-VAE forward ({H}x{W}x3, K={K}, D={D})",
            "value": round(B * n_gpus * args.steps / elapsed, 1), "unit": "images/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            # fp32 in/out and fp32 accumulation everywhere; the conv products of the whole-path kernels are formed from two
            # fp16 terms per operand (three term products, relative error <= 2^-21 per product: representation 2^-23 per
            # operand + the dropped pair 2^-22 -- fp32's own is 2^-24; VQVAE_CONV_BF16_SPLIT selects the exact three-term
            # bf16 splits, 3*2^-24, VQVAE_CONV_EXACT_FP32 the plain fp32-MFMA kernels); quantizer indices bit-exact for
            # identical z_e
            "dtype": "f32 (fp32 in/out/accumulate; conv products from two fp16 terms per operand, <= 2^-21 rel per product)"
                     if conv_backend == "hip" else "f32",
            "data": "synthetic",
            "config": {"workload": desc, "per_gpu_batch": B, "global_batch": B * n_gpus, "image": [3, H, W],
                       "K": K, "D": D, "parallelism": f"batch-sharded replicas x{n_gpus}, no collective"},
            "timing": {"repeats": len(times), "steps_per_repeat": args.steps, "statistic": "median repeat",
                       "ms_per_step_min": round(srt[0] / args.steps * 1e3, 4),
                       "ms_per_step_max": round(srt[-1] / args.steps * 1e3, 4),
                       "timed_seconds_total": round(sum(times), 3)},
        }
        line["per_rank"] = per_rank
        if calib:
            # next to `value`: the box's bare-MFMA rate and clock, and the value on the calibration's reference box
            line = {k: v for k, v in list(line.items())[:2]} | {
                "value_normalised": round(line["value"] * CALIB_REFERENCE_TFLOPS / max(calib["mfma_fp16_random_tflops"], 1.0), 1),
                "calibration_mfma_fp16_random_tflops": calib["mfma_fp16_random_tflops"], "calibration_sclk_ghz": calib["sclk_ghz"],
                "calibration_reference_tflops": CALIB_REFERENCE_TFLOPS} | {k: v for k, v in list(line.items())[2:]}
            line["calibration"] = calib
            # the same build on the calibration's reference box: value x (reference / measured) -- a first-order correction (the
            # step is matrix-bound: 93 % of it is conv kernels at 60-70 % of this stream's rate)
            line["value_normalised"] = round(line["value"] * CALIB_REFERENCE_TFLOPS / max(calib["mfma_fp16_random_tflops"], 1.0), 1)
        if dry:
            line["dry_run"] = True
            line["data"] = "dry-run stub (no kernels ran; not a measurement)"
        else:
            rows = B * (H // 4) * (W // 4)
            pmc, pmc_stale = pmc_traffic(args.workload, _lib.vq_kernel_name(K, D, 0x1))
            if vq_n:
                t_vq = vq_ms / vq_n * 1e-3                     # seconds per launch
                alg_bytes = rows * (8 * D + 8)                 # read z_e, write z_q, write int64 idx
                achieved = alg_bytes / t_vq / 1e9
                vq_inst = _lib.vq_kernel_instance(rows, K, D, (H // 4) * (W // 4), 0x1)
                line["roofline"] = {
                    "kernel": vq_inst + " (fused VQ: 16-bit MFMA screen with a rigorous bound + exact "
                              "fp32 refine of the surviving codes; bit-exact indices)",
                    "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "traffic": vq_instance_traffic(pmc, vq_inst, rows),
                    "traffic_over_algorithmic": (round(vq_instance_traffic(pmc, vq_inst, rows) / alg_bytes, 3)
                                                 if vq_instance_traffic(pmc, vq_inst, rows) else None),
                    "traffic_source": pmc["_path"] if vq_instance_traffic(pmc, vq_inst, rows) else None,
                    "traffic_stale": pmc_stale or (None if vq_instance_traffic(pmc, vq_inst, rows) or not pmc else
                                                   f"{pmc['_path']}: no PMC pass for {vq_inst} at {rows} rows"),
                    "avg_kernel_us": round(t_vq * 1e6, 2), "rows_per_launch": rows,
                    "in_step": "own launch" if vq_in_step else
                               "the step quantizes inside the encoder's last kernel (conv_res_pair8_h2_kernel<2, true>; z_e is never "
                               "written); these figures are the standalone kernel vqvae_vq_forward_f32 launches, timed on the same "
                               "batch's z_e in extra launches after the timed region",
                    "alg_bytes_per_row": 8 * D + 8,
                    "screen_tflops_16bit": round(2.0 * rows * K * D * _lib.vq_sweeps(K, D) / t_vq / 1e12, 1),
                    "hbm_achievable_frac": round(achieved / HBM_ACHIEVABLE_GBPS, 4),
                    "note": "algorithmic bytes = rows x (8D+8): read z_e, write z_q, write int64 idx; avg_kernel_us is "
                            "the live HIP-event average over the instrumented launches, the events being the dispatch's own start / "
                            "stop (hipExtLaunchKernelGGL) on the launch stream -- the interval rocprofv3's kernel trace reports; "
                            "two marker events around the launch read 2-3 us more; an exhaustive exact-fp32 sweep "
                            "(VQVAE_VQ_EXACT_SWEEP) is capped at 15.6% of HBM peak by arithmetic at K=512, D=64",
                }
            if conv_backend == "hip" and "conv_igemm" in extra and "res_layer" in extra:
                # 32x32 images: the first layer runs inside the encoder's second conv (enc_front8_h2_kernel, "conv_igemm") and
                # the last inside the decoder's tail kernel (dec_tail8_h2_kernel, "conv_out"): every conv kernel counts then
                ends = "conv_in" not in extra and "conv_out" in extra
                t_conv = (extra["conv_igemm"]["ms_per_step"] + extra["res_layer"]["ms_per_step"] +
                          (extra["conv_out"]["ms_per_step"] if ends else 0.0)) * 1e-3
                flops_img = conv_flops_per_image(H, W, D, ends=ends)
                alg_tf = B * flops_img / t_conv / 1e12
                # 16-bit MFMA term products per fp32 multiply-add on this workload's maps (3 = two-term fp16 on 8x8 maps,
                # 6 = three-term bf16 on larger ones), asked of the library for the 3x3 layer at the latent resolution
                terms = _lib.load().vqvae_conv_term_products(1, H // 4, W // 4, 128, 128, 0x100) or 6   # 0x100: as the whole path launches it
                scheme = ("two-term fp16 products: 3 fp16 MFMA term products per fp32 product" if terms == 3 else
                          "three-term bf16 products: 6 bf16 MFMA term products per fp32 product")
                line["roofline_conv"] = {
                    "kernel": ("all four conv kernels of the step: encoder front (first two layers), encoder 3x3 + residual stack + "
                               "pre-quantisation 1x1, decoder conv-transpose 3x3 + residual stack, decoder tail (last two layers) "
                               if ends else
                               "conv / conv-transpose / fused residual-layer kernels between the first and the last layer ") +
                              f"({scheme})",
                    "bound": "mfma", "achieved": round(terms * alg_tf, 1), "peak": MFMA_16BIT_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(terms * alg_tf / MFMA_16BIT_PEAK_TFLOPS, 4),
                    "term_products_per_mac": terms,
                    "achieved_algorithmic_tflops": round(alg_tf, 1),
                    "frac_of_split_ceiling": round(alg_tf / (MFMA_16BIT_PEAK_TFLOPS / terms), 4),
                    "frac_of_fp32_mfma_peak": round(alg_tf / MFMA_F32_PEAK_TFLOPS, 4),
                    "traffic": int(pmc["conv_bytes_per_image"] * B) if pmc and "conv_bytes_per_image" in pmc else None,
                    "traffic_source": pmc["_path"] if pmc and "conv_bytes_per_image" in pmc else None,
                    "flops_per_image": flops_img, "ms_per_step": round(t_conv * 1e3, 4),
                    # the same flop count over the TIMED region's step (no event brackets; on the fused 32x32 path the step is
                    # these kernels + a 5 us finalize, the quantizer's work inside the second one included in the time)
                    "frac_over_timed_step": (round(terms * B * flops_img / (elapsed / args.steps) / 1e12 / MFMA_16BIT_PEAK_TFLOPS, 4)
                                             if ends and vq_in_step is False else None),
                    "note": f"achieved = 16-bit MFMA flop ISSUED ({terms} term products per fp32 product) / live HIP-event "
                            "time; achieved_algorithmic_tflops = 2*MAC of the layers / the same time (what a plain fp32 conv "
                            f"would be credited with): its ceiling on this path is 2500/{terms} = {2500 // terms} TF",
                }
            # the library's four timing slots, named by what runs in them on this workload
            fusedp = conv_backend == "hip" and "conv_in" not in extra and "conv_out" in extra
            slot_names = ({"conv_igemm": "enc_front8_h2_kernel",
                           "res_layer": "conv_res_pair8_h2_kernel<2> + <0>" if vq_in_step else
                                        "conv_res_pair8_h2_kernel<2, true> (encoder 3x3 + residual stack + 1x1 + QUANTIZER) + <0>",
                           "conv_out": "dec_tail8_h2_kernel"} if fusedp else
                          {"conv_igemm": "conv_igemm_bf3 / conv_tile8_bf3 kernels", "res_layer": "res_layer / res_pair kernels",
                           "conv_in": "conv_in kernels", "conv_out": "convt_out_kernel"})
            line["kernels"] = {slot_names.get(k, k): dict(v, timing="hip-event brackets (event_inflated: the brackets add "
                                                                   "~10 % to a step; profiles/ holds the rocprofv3 figures)")
                               for k, v in extra.items()}
            line["source_sha"] = source_sha()
            if n_gpus == 1 and not args.no_power:
                # package power / shader clock while the SAME step runs back to back (its own ~3 s run behind the timed region: the
                # hwmon files of this GPU, 50 ms apart).  The step sits at the package limit -- that, not issue slots, is its ceiling
                try:
                    line["power"] = step_power(step, torch, dev)
                except Exception as e:                 # noqa: BLE001
                    line["power"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            if n_gpus == 1 and not args.no_cpu_baseline:
                if args.workload == "c3" and conv_backend == "hip":
                    line["index_flips_vs_reference"] = index_flips(model, x, torch)
                line["cpu_baseline"] = cpu_baseline(args.cpu_seconds, torch)
            if n_gpus == 1 and args.workload == "c3" and not args.no_other_workloads and not args.batch:
                del out
                # side figures of the line -- the other BASELINE configs, the headline's step in the two exacter product schemes (the
                # headline itself = "fp16x2"), main.py's other hyper-parameters, the wire format, the quantizer's size sweep: none of
                # them may take the headline line down with it (an error is reported in its place, not swallowed)
                legs = [(w, (lambda w=w: other_workload(w, torch, dev))) for w in ("c2", "c4", "c5")]
                legs += [("c3_bf16x3", lambda: scheme_leg("bf16x3", torch, dev)), ("c3_fp32", lambda: scheme_leg("fp32", torch, dev))]
                legs += [("c3_trained", lambda: trained_leg("trained_main_defaults", torch, dev)),
                         ("c3_trained_b128x20k", lambda: trained_leg("trained_b128x20k", torch, dev, seconds=0.6))]
                legs += [(name, (lambda name=name: hparam_leg(name, torch, dev, seconds=0.6))) for name in HPARAM_LEGS]
                legs += [("wire_format", lambda: wire_legs(torch, dev, elapsed / args.steps * 1e3)),
                         ("vq_sizes", lambda: vq_size_sweep(torch, dev))]
                line["other_workloads"] = {}
                for name, fn in legs:
                    try:
                        line["other_workloads"][name] = fn()
                    except Exception as e:             # noqa: BLE001
                        line["other_workloads"][name] = {"error": f"{type(e).__name__}: {e}"[:300]}
                        torch.cuda.empty_cache()
                try:                                   # a next-row figure: its failure must not take the headline line with it
                    line["training_step"] = training_step(torch, dev)
                except Exception as e:                 # noqa: BLE001  (reported in the line, not swallowed)
                    line["training_step"] = {"error": f"{type(e).__name__}: {e}"}
                try:
                    line["pixelcnn"] = pixelcnn_leg(torch, dev)
                except Exception as e:                 # noqa: BLE001
                    line["pixelcnn"] = {"error": f"{type(e).__name__}: {e}"}
                conv_mod.set_conv_backend(conv_backend)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
