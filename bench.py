#!/usr/bin/env python3
"""bench.py -- images/sec of the VQ-VAE forward (32x32x3, K=512, D=64) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2] [--batch B]

One process per GPU (the driver launches N>1 through torch.distributed.run).  The
batch shards embarrassingly: every rank runs the same per-GPU batch on its own
replica of the weights (weak scaling), there is no data-path collective; RCCL is
used only for the barrier and the max-over-ranks of the elapsed time.

A "step" is one `VQVAE.forward(x)` over one synthetic batch already resident in HBM:
  c3 (default)  BASELINE config 3: full HIP path (Encoder + VQ + Decoder kernels), B=4096/GPU
  c2            BASELINE config 2: HIP VectorQuantizer, torch convs unchanged,      B=1024/GPU

Rank 0 prints ONE JSON line.  `roofline` is for the fused VectorQuantizer kernel
(timed live with HIP events on the launch stream, vqvae_profile_* hooks, in extra
steps after the timed region); `cpu_baseline` is the reference's algorithm on the
host cores (oracle/torch_port.py, same ATen ops as the reference, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

METRIC = "images/sec VQ-VAE forward (32x32x3, K=512, D=64)"
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # exact-fp32 MFMA = vector rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="auto", choices=["auto", "c3", "c2"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    return ap.parse_args()


def cpu_baseline(seconds: float):
    """Reference algorithm on the host: same ATen CPU ops in the same order as the reference
    (oracle/torch_port.py), main.py defaults, eval + no_grad, on a bounded sample.  A few thread
    counts are tried briefly (big hosts thrash on 32-image batches) and the best is reported with
    the thread count that produced it."""
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
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(best, 1), "unit": "images/s", "cores": best_t, "kind": "port",
            "sample": "VQVAE.forward 32x32x3 K=512 D=64 fp32 eval/no_grad on the host CPU (" + cpu +
                      f", {ncpu} logical cpus); ~{per:.1f}s per (threads,batch) point, img/s: " + " ".join(detail)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the HIP path has no CPU fallback")
    # dry-run switches for a 1-GPU box (exercise the N>1 control flow without N GPUs): all ranks on device 0
    # and gloo for the barrier / max-reduce.  The driver's real runs use neither.
    share_gpu = os.environ.get("VQVAE_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("VQVAE_BENCH_DIST_BACKEND", "nccl")
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist_mod.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_mod
    n_gpus = world

    from vqvae_amd import _lib, conv
    from vqvae_amd.modules import VQVAE
    _lib.load()                                    # fail loudly if the HIP library is missing

    workload = args.workload
    if workload == "auto":
        try:
            from vqvae_amd import conv_hip  # noqa: F401
            workload = "c3"
        except ImportError:
            workload = "c2"
    conv.set_conv_backend("hip" if workload == "c3" else "torch")
    B = args.batch or (4096 if workload == "c3" else 1024)
    K, D, H, W = 512, 64, 32, 32

    torch.manual_seed(0)                           # same weights on every rank (replicas)
    model = VQVAE(128, 32, 2, K, D, 0.25).eval().to(dev)
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(B, 3, H, W, generator=g).to(dev)   # synthetic, resident in HBM before timing

    def step():
        with torch.no_grad():
            return model(x)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out[1]).all()

    # ---- per-kernel time of the fused VQ kernel, extra steps outside the timed region ----
    _lib.profile_enable(True)
    nprof = min(args.steps, 50)
    for _ in range(nprof):
        step()
    vq_ms, vq_n = _lib.profile_collect("vq_main")
    extra = {}
    for name in ("conv_igemm", "res_layer", "conv_in", "conv_out"):
        ms, n = _lib.profile_collect(name)
        if n:
            extra[name] = {"ms_per_step": round(ms / nprof, 4), "launches_per_step": n // nprof}
    _lib.profile_enable(False)

    if rank == 0:
        rows = B * (H // 4) * (W // 4)
        t_vq = vq_ms / max(vq_n, 1) * 1e-3            # seconds per launch
        alg_bytes = rows * (8 * D + 8)                 # read z_e, write z_q, write int64 idx
        achieved = alg_bytes / t_vq / 1e9
        flops = 2.0 * rows * K * D
        roofline = {
            "kernel": "vq_filter_kernel_d64 (fused VQ: bf16-MFMA screen with a rigorous bound + exact fp32 "
                      "refine of the surviving codes; bit-exact indices)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
            # measured offline on a 4.19 M-row stream: 520.7 B/row (profiles/r01_vq_hbm_traffic.txt)
            "traffic": rows * 521,
            "avg_kernel_us": round(t_vq * 1e6, 2), "rows_per_launch": rows,
            "alg_bytes_per_row": 8 * D + 8,
            "screen_tflops_bf16": round(2 * flops / t_vq / 1e12, 1),
            "hbm_achievable_frac": round(achieved / 6290.0, 4),
            "note": "algorithmic bytes = rows x (8D+8): read z_e, write z_q, write int64 idx; the screen "
                    "sweeps the codebook twice on the bf16 matrix cores (2 x 2KD flop/row); an exhaustive "
                    "exact-fp32 sweep (VQVAE_VQ_EXACT_SWEEP) is capped at 15.6% of HBM peak by arithmetic",
        }
        # the convs dominate the step time: their matrix-pipe roofline next to the quantizer's HBM one.
        # MAC*2 per image at 32x32: enc 4x4s2 16.8 M + enc 3x3 18.9 M + 1x1 1.05 M + dec convT3x3
        # 9.4 M + dec convT4x4s2 16.8 M + 4 residual layers 21.0 M = 83.9 MF on the split-bf16 path (6 bf16 MFMA
        # term products per fp32 product); the first/last layers (3.1 MF, also split-bf16) are memory-bound and
        # reported under "kernels" only
        roofline_conv = None
        if workload == "c3" and "conv_igemm" in extra and "res_layer" in extra:
            t_conv = (extra["conv_igemm"]["ms_per_step"] + extra["res_layer"]["ms_per_step"]) * 1e-3
            bf16_tf = B * 83.9e6 * 6 / t_conv / 1e12
            roofline_conv = {
                "kernel": "conv_tile8_bf3_kernel + res_tile8_bf3_kernel (9 launches per step, split-bf16 products)",
                "bound": "mfma", "achieved": round(bf16_tf, 1), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(bf16_tf / 2500.0, 4),
                # HBM-side bytes per step of these nine launches from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
                # offline at B=4096: 815.7 kB per image = 1.31x the 622.6 kB algorithmic; profiles/r01_c3_hbm_traffic_v8.txt)
                "traffic": int(B * 815.7e3),
                "fp32_equivalent_tflops": round(bf16_tf / 6, 1), "ms_per_step": round(t_conv * 1e3, 4),
                "note": "achieved = bf16 MFMA flop issued (6 term products per fp32 product) / live HIP-event time of "
                        "those kernels; the matrix pipe sustains ~1900 TF with random operands "
                        "(tools/ubench/mfma_bf16_peak.hip), 2500 TF is the dense spec peak at 2.4 GHz; PMC: the chip "
                        "holds ~2.0 GHz under this load and the pipe is 59-60% busy in the conv kernels, 43% in the "
                        "residual layers (profiles/r01_c3_pmc_util_v7.txt)",
            }
        line = {
            "metric": METRIC, "value": round(B * n_gpus * args.steps / elapsed, 1), "unit": "images/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            # fp32 in/out and fp32 accumulation everywhere; conv products are formed from exact
            # three-term bf16 splits of the fp32 operands (error <= 3*2^-24 per product, fp32-grade;
            # VQVAE_CONV_EXACT_FP32 selects the plain fp32-MFMA kernels); quantizer indices bit-exact
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"c3": "BASELINE config 3: full HIP path (Encoder+VQ+Decoder), "
                                          "32x32x3, K=512, D=64",
                                    "c2": "BASELINE config 2: HIP VectorQuantizer kernel, torch (MIOpen) "
                                          "convs unchanged, 32x32x3, K=512, D=64"}[workload],
                       "per_gpu_batch": B, "global_batch": B * n_gpus, "image": [3, H, W],
                       "K": K, "D": D, "parallelism": f"batch-sharded replicas x{n_gpus}, no collective"},
            "roofline": roofline,
            "roofline_conv": roofline_conv,
            "kernels": extra,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
