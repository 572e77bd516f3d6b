"""Structured synthetic 32x32x3 images  --  TEST / TOOL INFRASTRUCTURE (no dataset can be downloaded here).

Stand-in for the CIFAR-10 loader of the reference (`utils.py:11-27`): images in [0, 1] made of a low-frequency colour field,
one to three solid shapes (boxes, discs), an oriented stripe texture on some of them and a little pixel noise, then
normalised exactly as `utils.py:15-16` does (`Normalize((0.5,)*3, (0.5,)*3)` -> [-1, 1]).  Unlike N(0, 1) noise these have
spatial structure an encoder can compress, so a model trained on them (tools/train_checkpoint.py, main.py:67-98's loop)
ends up with the per-channel weight statistics and the code usage of a really trained checkpoint.

Deterministic for a given (n, seed) on the CPU generator of one torch version; the GPU box runs the same image.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def images01(n: int, seed: int, size: int = 32) -> torch.Tensor:
    """-> (n, 3, size, size) fp32 in [0, 1]"""
    g = torch.Generator().manual_seed(seed)
    S = size
    # low-frequency colour field: 4x4 control points per channel, correlated across channels, bicubic to SxS
    base = torch.rand(n, 1, 4, 4, generator=g)
    tint = 0.35 * torch.randn(n, 3, 4, 4, generator=g)
    field = F.interpolate(base + tint, size=(S, S), mode="bicubic", align_corners=False)
    mean = torch.rand(n, 3, 1, 1, generator=g)
    img = 0.55 * field + 0.45 * mean
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing="ij")
    yy, xx = yy.view(1, S, S), xx.view(1, S, S)
    # stripes on ~40 % of the images: random orientation, period 2.5 ... 10 pixels, inside a soft window
    has = (torch.rand(n, generator=g) < 0.4).float().view(n, 1, 1)
    th = torch.rand(n, generator=g).view(n, 1, 1) * math.pi
    per = (2.5 + 7.5 * torch.rand(n, generator=g)).view(n, 1, 1)
    ph = torch.rand(n, generator=g).view(n, 1, 1) * 2 * math.pi
    amp = (0.08 + 0.2 * torch.rand(n, generator=g)).view(n, 1, 1)
    stripes = torch.sin(2 * math.pi * (xx * torch.cos(th) + yy * torch.sin(th)) / per + ph)
    cx0 = torch.rand(n, generator=g).view(n, 1, 1) * S
    cy0 = torch.rand(n, generator=g).view(n, 1, 1) * S
    win = torch.exp(-((xx - cx0) ** 2 + (yy - cy0) ** 2) / (2 * (0.35 * S) ** 2))
    img = img + (has * amp * stripes * win).unsqueeze(1) * (0.5 + torch.rand(n, 3, 1, 1, generator=g))
    # one to three solid shapes
    for s in range(3):
        on = (torch.rand(n, generator=g) < (1.0, 0.6, 0.3)[s]).view(n, 1, 1)
        disc = (torch.rand(n, generator=g) < 0.5).view(n, 1, 1)
        cx = (torch.rand(n, generator=g) * S).view(n, 1, 1)
        cy = (torch.rand(n, generator=g) * S).view(n, 1, 1)
        rx = (2 + torch.rand(n, generator=g) * 0.3 * S).view(n, 1, 1)
        ry = (2 + torch.rand(n, generator=g) * 0.3 * S).view(n, 1, 1)
        inside_box = ((xx - cx).abs() <= rx) & ((yy - cy).abs() <= ry)
        inside_disc = ((xx - cx) / rx) ** 2 + ((yy - cy) / rx) ** 2 <= 1.0
        mask = (torch.where(disc, inside_disc, inside_box) & on).unsqueeze(1)
        colour = torch.rand(n, 3, 1, 1, generator=g)
        shade = 1.0 + 0.15 * ((xx - cx) / S).unsqueeze(1)                 # a little shading across the shape
        img = torch.where(mask, (colour * shade).expand(n, 3, S, S), img)
    img = img + 0.02 * torch.randn(n, 3, S, S, generator=g)
    return img.clamp_(0.0, 1.0).contiguous()


def normalised(n: int, seed: int, size: int = 32) -> torch.Tensor:
    """utils.py:15-16: ToTensor() then Normalize(0.5, 0.5) -> [-1, 1]"""
    return ((images01(n, seed, size) - 0.5) / 0.5).contiguous()


def train_var(x01: torch.Tensor) -> float:
    """utils.py:86 `x_train_var = np.var(training_data.train_data / 255.0)`: the variance of the [0, 1] pixels"""
    return float(x01.double().var(unbiased=False))
