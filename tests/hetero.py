"""Trained-checkpoint-like weight distributions for the parity tests ("parity is pinned on friendly data only").

A freshly initialised VQVAE has ONE magnitude per layer by construction (Kaiming-uniform), which is the friendliest
possible input for a product scheme that carries one power-of-two scale per weight tensor.  A trained checkpoint
(`main.py:74-79`, stored through `utils.py:109-113`) has per-channel weight norms that differ by orders of magnitude.
Two generators, both starting from the reference's default init (same keys, same aliasing of the residual layers):

* `rescale_coupled`  -- a diagonal re-parametrisation of the network: the output channels of every layer are
  multiplied by 10^U(-d, d) and the input channels of the layer behind it by the reciprocals (ReLU commutes with positive
  channel factors; the residual stream keeps one factor vector through the stack, as the shared weights demand).  In exact
  arithmetic the function is the default model's up to the channel factors of z_e / x_hat, so activations stay in range
  whatever the depth -- and EVERY term of a reduction carries the same weight in the result while its two operands span
  2d decades: the worst case for per-tensor operand scales.
* `rescale_independent` -- input AND output channels of every weight by independent 10^U(-d, d) factors (biases by
  their channel's factor), then every layer divided by a scalar so that its output stays at unit magnitude on a
  calibration batch (run through the oracle): reductions dominated by a few channels, tiny terms beside huge ones.

Both return a fresh state_dict in the reference's layout; the codebook is replaced by rows of the oracle's z_e on a
calibration batch plus noise (a trained codebook lives where z_e lives), so that index comparisons mean something.
"""
from __future__ import annotations

import torch

E = "encoder.conv_stack."
Dk = "decoder.inverse_conv_stack."
RES = ("res_block.1.weight", "res_block.3.weight")


def _factors(g, n, decades):
    return 10.0 ** ((torch.rand(n, generator=g) * 2 - 1) * decades)


def _alias(sd, n_res):
    for pre in (E + "5.stack.", Dk + "1.stack."):
        for l in range(1, n_res):
            for blk in RES:
                sd[f"{pre}{l}.{blk}"] = sd[pre + "0." + blk]
    return sd


def _trained_like_codebook(sd, n_res, g, K, D, calib=None):
    from oracle import torch_port
    x = torch.randn(max(64, -(-K // 32)), 3, 32, 32, generator=g) if calib is None else calib
    with torch.no_grad():
        z = torch_port.encode(sd, x.clone(), n_res).permute(0, 2, 3, 1).reshape(-1, D)
    sel = torch.randperm(z.shape[0], generator=g)[:K]
    cb = z[sel].clone()
    cb += 0.05 * cb.abs().mean(0, keepdim=True) * torch.randn(K, D, generator=g)
    return cb.contiguous()


def rescale_coupled(sd0, seed, decades=3.0, n_res=2, codebook=True):
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.detach().clone() for k, v in sd0.items()}
    c0, c1 = sd[E + "0.weight"].shape[0], sd[E + "2.weight"].shape[0]
    rh = sd[E + "5.stack.0." + RES[0]].shape[0]
    K, D = sd["vector_quantization.embedding.weight"].shape
    f0, f2, s, fh, fz = (_factors(g, n, decades) for n in (c0, c1, c1, rh, D))
    sd_, fhd, f2d, fo = (_factors(g, n, decades) for n in (c1, rh, c0, 3))

    def conv(key, fout, fin, bias=True):                       # Conv2d weight (Cout, Cin, kh, kw)
        sd[key + ".weight"] = sd[key + ".weight"] * fout.view(-1, 1, 1, 1) / fin.view(1, -1, 1, 1)
        if bias:
            sd[key + ".bias"] = sd[key + ".bias"] * fout

    def convt(key, fout, fin):                                 # ConvTranspose2d weight (Cin, Cout, kh, kw)
        sd[key + ".weight"] = sd[key + ".weight"] * fout.view(1, -1, 1, 1) / fin.view(-1, 1, 1, 1)
        sd[key + ".bias"] = sd[key + ".bias"] * fout

    one3 = torch.ones(3)
    conv(E + "0", f0, one3)
    conv(E + "2", f2, f0)
    conv(E + "4", s, f2)
    conv(E + "5.stack.0.res_block.1", fh, s, bias=False)
    conv(E + "5.stack.0.res_block.3", s, fh, bias=False)
    conv("pre_quantization_conv", fz, s)
    sd["vector_quantization.embedding.weight"] = sd["vector_quantization.embedding.weight"] * fz.view(1, -1)
    convt(Dk + "0", sd_, fz)
    conv(Dk + "1.stack.0.res_block.1", fhd, sd_, bias=False)
    conv(Dk + "1.stack.0.res_block.3", sd_, fhd, bias=False)
    convt(Dk + "2", f2d, sd_)
    convt(Dk + "4", fo, f2d)
    _alias(sd, n_res)
    if codebook:
        sd["vector_quantization.embedding.weight"] = _trained_like_codebook(sd, n_res, g, K, D)
    return {k: v.contiguous() for k, v in sd.items()}


def rescale_independent(sd0, seed, decades=3.0, n_res=2, calib_batch=16):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.detach().clone() for k, v in sd0.items()}
    K, D = sd["vector_quantization.embedding.weight"].shape

    def scale(key, transposed, bias):
        w = sd[key + ".weight"]
        co, ci = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
        fo, fi = _factors(g, co, decades), _factors(g, ci, decades)
        if transposed:
            sd[key + ".weight"] = w * fi.view(-1, 1, 1, 1) * fo.view(1, -1, 1, 1)
        else:
            sd[key + ".weight"] = w * fo.view(-1, 1, 1, 1) * fi.view(1, -1, 1, 1)
        if bias:
            sd[key + ".bias"] = sd[key + ".bias"] * fo

    def norm(key, t, bias):                                    # divide the layer by its output's magnitude
        m = float(t.abs().max())
        m = m if m > 0 else 1.0
        sd[key + ".weight"] = sd[key + ".weight"] / m
        if bias:
            sd[key + ".bias"] = sd[key + ".bias"] / m
        return t / m

    def res_stack(t, pre):
        # the two layers share weights: factors once, normalised on the first application
        k1, k3 = pre + "res_block.1", pre + "res_block.3"
        scale(k1, False, False)
        scale(k3, False, False)
        r = F.relu(t)
        h = norm(k1, F.conv2d(r, sd[k1 + ".weight"], None, 1, 1), False)
        norm(k3, F.conv2d(F.relu(h), sd[k3 + ".weight"], None, 1, 0), False)
        for _ in range(n_res):
            r = F.relu(t)
            t = r + F.conv2d(F.relu(F.conv2d(r, sd[k1 + ".weight"], None, 1, 1)), sd[k3 + ".weight"], None, 1, 0)
        return F.relu(t)

    with torch.no_grad():
        x = torch.randn(calib_batch, 3, 32, 32, generator=g)
        t = x
        for key, st in ((E + "0", 2), (E + "2", 2), (E + "4", 1)):
            scale(key, False, True)
            t = norm(key, F.conv2d(t, sd[key + ".weight"], sd[key + ".bias"], st, 1), True)
            if st == 2:
                t = F.relu(t)
        t = res_stack(t, E + "5.stack.0.")
        scale("pre_quantization_conv", False, True)
        z = norm("pre_quantization_conv", F.conv2d(t, sd["pre_quantization_conv.weight"], sd["pre_quantization_conv.bias"]), True)
        _alias(sd, n_res)
        sd["vector_quantization.embedding.weight"] = _trained_like_codebook(sd, n_res, g, K, D)
        t = z
        scale(Dk + "0", True, True)
        t = norm(Dk + "0", F.conv_transpose2d(t, sd[Dk + "0.weight"], sd[Dk + "0.bias"], 1, 1), True)
        t = res_stack(t, Dk + "1.stack.0.")
        scale(Dk + "2", True, True)
        t = F.relu(norm(Dk + "2", F.conv_transpose2d(t, sd[Dk + "2.weight"], sd[Dk + "2.bias"], 2, 1), True))
        scale(Dk + "4", True, True)
        norm(Dk + "4", F.conv_transpose2d(t, sd[Dk + "4.weight"], sd[Dk + "4.bias"], 2, 1), True)
    _alias(sd, n_res)
    return {k: v.contiguous() for k, v in sd.items()}


def outlier_images(B, seed, kind):
    """N(0,1) images with in-image outliers: one pixel (all three channels) or one whole channel 1e4 x the rest."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, 32, 32, generator=g)
    for b in range(B):
        if kind == "pixel":
            yy, xx = int(torch.randint(0, 32, (1,), generator=g)), int(torch.randint(0, 32, (1,), generator=g))
            x[b, :, yy, xx] *= 1.0e4
        elif kind == "channel":
            x[b, b % 3] *= 1.0e4
        elif kind == "mixed":
            if b % 3 == 0:
                x[b, :, (7 * b) % 32, (11 * b) % 32] *= 1.0e4
            elif b % 3 == 1:
                x[b, b % 2] *= 1.0e4
    return x


def per_channel_check(got, ref, what, atol_rel=1e-5, rtol=1e-4, per_image=True):
    """|got - ref| <= atol_rel * max|ref over the (image,) channel| + rtol * |ref|, for NCHW arrays; returns the worst
    error in units of that channel maximum (for the printed report)."""
    import numpy as np
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    ax = (2, 3) if per_image else (0, 2, 3)
    cmax = np.abs(ref).max(axis=ax, keepdims=True)
    err = np.abs(got - ref)
    lim = atol_rel * cmax + rtol * np.abs(ref)
    bad = err > lim
    worst = float((err / np.maximum(cmax, 1e-300)).max())
    if bad.any():
        i = np.unravel_index(np.argmax(err / np.maximum(lim, 1e-300)), err.shape)
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} elements outside atol {atol_rel:g} x channel max + rtol "
                             f"{rtol:g}; worst at {i}: got {got[i]:.9g} ref {ref[i]:.9g} (channel max {cmax[i[0] if per_image else 0, i[1], 0, 0]:.3g}); "
                             f"largest error = {worst:.3g} x its channel's maximum")
    return worst
