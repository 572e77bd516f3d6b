"""Seeded input generators shared by oracle/gen_golden.py (which runs the real
reference on them, in the build container) and by the tests (which regenerate
the same inputs anywhere and compare against the committed outputs).

torch's CPU generator is deterministic for a given torch version, and the GPU
box runs the same image, so only OUTPUTS (and input hashes) are committed.
"""
from __future__ import annotations

import hashlib

import numpy as np
import torch

# name -> (K, D, B, H, W, beta, kind)
VQ_CASES = {
    # main.py defaults at BASELINE config-1 size (N = 2048 rows)
    "k512_d64_c1":      (512, 64, 32, 8, 8, 0.25, "init"),
    # codebook and z both unit normal: wide margins, every code used
    "k512_d64_normal":  (512, 64, 8, 8, 8, 0.25, "normal"),
    # BASELINE config-4 codebook (two LDS chunks), non-power-of-two grid
    "k1024_d64":        (1024, 64, 2, 14, 14, 0.25, "init"),
    # BASELINE config-5 codebook (K=8192, D=128)
    "k8192_d128":       (8192, 128, 1, 8, 8, 0.25, "init"),
    # small / ragged shapes: K not a multiple of 32, H != W, N not a multiple of 32
    "k100_d32_ragged":  (100, 32, 3, 5, 7, 0.5, "normal"),
    "k64_d256":         (64, 256, 2, 4, 4, 0.25, "normal"),
    # exact ties: duplicated codebook rows and z rows equal to codes -> first index must win
    "ties":             (96, 64, 2, 8, 8, 0.25, "ties"),
    # NaN / Inf rows: torch.argmin treats NaN as minimal (first NaN wins)
    "nonfinite":        (64, 64, 1, 8, 8, 0.25, "nonfinite"),
    # round 5: embedding widths outside {32, 64, 128, 256} (main.py:21 leaves --embedding_dim free): the exact-fp32 vector kernel.
    # 48 = the width VERDICT r4 names; 7 = no 16-byte pieces and ATen's scalar row sum; 200 = ragged vector tail; ties / non-finite rows
    # (D <= 256: beyond 383 the reference's own matmul stops being one fmaf chain -- MKL blocks the reduction)
    "k96_d48_ragged":   (96, 48, 3, 5, 7, 0.25, "normal"),
    "k300_d48_init":    (300, 48, 2, 8, 8, 0.25, "init"),
    "k50_d7":           (50, 7, 2, 3, 5, 0.5, "normal"),
    "k40_d200":         (40, 200, 1, 4, 4, 0.25, "normal"),
    "ties_d48":         (96, 48, 2, 8, 8, 0.25, "ties"),
    "nonfinite_d72":    (64, 72, 1, 8, 8, 0.25, "nonfinite"),
}


def sha(a) -> str:
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().contiguous().numpy()
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def vq_inputs(name):
    """-> (z_e (B,D,H,W) fp32, codebook (K,D) fp32, beta)"""
    K, D, B, H, W, beta, kind = VQ_CASES[name]
    g = torch.Generator().manual_seed(1234 + sum(map(ord, name)))
    if kind == "init":
        # reference init: codebook ~ U(-1/K, 1/K) (models/quantizer.py:27); z at the
        # scale a random-init encoder produces (||z||^2 ~ 0.28)
        cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
        z = torch.randn(B, D, H, W, generator=g) * 0.066
    elif kind == "normal":
        cb = torch.randn(K, D, generator=g)
        z = torch.randn(B, D, H, W, generator=g)
    elif kind == "ties":
        cb = torch.randn(K, D, generator=g)
        cb[K // 2:] = cb[:K - K // 2]                     # every code appears twice
        cb[7] = cb[3]
        sel = torch.randint(0, K, (B * H * W,), generator=g)
        zr = cb[sel].clone()
        zr[::3] += 0.5 * torch.randn(zr[::3].shape, generator=g)
        zr[1::5] = 0.0                                    # all-zero rows
        z = zr.view(B, H, W, D).permute(0, 3, 1, 2).contiguous()
    elif kind == "nonfinite":
        cb = torch.randn(K, D, generator=g)
        zr = torch.randn(B * H * W, D, generator=g)
        zr[3, 5] = float("nan")
        zr[10, 0] = float("inf")
        zr[11, 63] = float("-inf")
        zr[12, :] = 3.0e19                                # zz overflows to +inf
        zr[13, 1] = 1.0e30
        zr[20, 7] = float("nan"); zr[20, 9] = float("inf")
        z = zr.view(B, H, W, D).permute(0, 3, 1, 2).contiguous()
    else:
        raise KeyError(kind)
    return z.float().contiguous(), cb.float().contiguous(), beta


# name -> (h_dim, res_h_dim, n_res_layers, K, D, beta, B, H, W)
MODEL_CASES = {
    # KAT-1 of SURVEY.md appendix B: main.py defaults, BASELINE config 1
    "kat1":      (128, 32, 2, 512, 64, 0.25, 32, 32, 32),
    # small, non-square, 3 residual layers, K not a power of two
    "small":     (64, 16, 3, 96, 32, 0.25, 3, 16, 24),
    # round 5: --embedding_dim 48 (the quantizer's exact-fp32 vector kernel inside the whole model)
    "d48":       (128, 32, 2, 512, 48, 0.25, 4, 32, 32),
}


def model_inputs(name):
    """x drawn from the same generator right after model construction with
    torch.manual_seed(0), exactly as KAT-1 prescribes.  Call AFTER building the model."""
    h, rh, nl, K, D, beta, B, H, W = MODEL_CASES[name]
    return torch.randn(B, 3, H, W)


# Round 6: really trained checkpoints.  name -> (h_dim, res_h_dim, n_res_layers, K, D, beta, B of the golden run, seed of its images).
# The weights come from tools/train_checkpoint.py (the loop of main.py:67-98 -- Adam amsgrad lr 3e-4, recon / x_train_var +
# embedding loss -- on the HIP training path, structured synthetic 32x32x3 images of tests/synthdata.py; logs in profiles/r06_train_*):
#   trained_main_defaults   main.py's own hyperparameters: batch 32, 5 000 updates  (perplexity 2.7 -> 42.5 over the run)
#   trained_b128x20k        batch 128, 20 000 updates                               (perplexity -> 114.7, 134 codes in use)
TRAINED_CASES = {
    "trained_main_defaults": (128, 32, 2, 512, 64, 0.25, 32, 4242),
    "trained_b128x20k":      (128, 32, 2, 512, 64, 0.25, 32, 4243),
}


def trained_state(name):
    """-> state_dict (reference layout, the aliased residual keys sharing storage again) of a committed trained checkpoint"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{name}_state.npz"))
    sd = {k: torch.from_numpy(z[k].copy()) for k in z.files}
    for pre in ("encoder.conv_stack.5.stack.", "decoder.inverse_conv_stack.1.stack."):
        l = 1
        while f"{pre}{l}.res_block.1.weight" in sd:
            for blk in ("res_block.1.weight", "res_block.3.weight"):
                assert torch.equal(sd[f"{pre}{l}.{blk}"], sd[f"{pre}0.{blk}"]), "aliased residual weights diverged in the checkpoint"
                sd[f"{pre}{l}.{blk}"] = sd[f"{pre}0.{blk}"]
            l += 1
    return sd
