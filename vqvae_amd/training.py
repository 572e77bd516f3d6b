"""Training-step companions of the forward path.

  * `vq_backward`           the VectorQuantizer gradients on the GPU (vqvae_vq_backward_f32)
  * `VQStraightThrough`     autograd.Function pairing the fused HIP forward with that backward, so
                            `main.py:74-79` (loss.backward()) works with the HIP quantizer; the convs
                            must then run on the "torch" backend (their backward is torch's)
  * `step_losses`           `recon_loss`, `loss` and `perplexity` of main.py:75-76,81-83 in one fused
                            reduction, returned as ONE 3-element device tensor (one D2H copy per step)

No CPU path and no fallback: CPU tensors raise VqvaeHipError.
"""
from __future__ import annotations

import torch

from . import _lib
from . import functional as F_hip


def _sp(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def vq_backward(z_e, codebook, idx, grad_zq, grad_loss, beta, *, rowmajor=False, need_z=True, need_codebook=True):
    """Gradients of VectorQuantizer.forward (models/quantizer.py:63-67) w.r.t. z_e and the codebook.

    z_e / grad_zq: (B,D,H,W), or (B,H,W,D) when rowmajor.  grad_loss: 0-dim device tensor or None (=1).
    Returns (grad_z or None, grad_codebook or None)."""
    F_hip._check_dev("z_e", z_e)
    F_hip._check_dev("codebook", codebook)
    F_hip._check_dev("idx", idx, torch.int64)
    z_e = z_e.contiguous()
    codebook = codebook.contiguous()
    idx = idx.contiguous()
    if rowmajor:
        B, H, W, D = z_e.shape
    else:
        B, D, H, W = z_e.shape
    K = codebook.shape[0]
    if codebook.shape[1] != D or idx.numel() != B * H * W:
        raise ValueError("shape mismatch between z_e, codebook and idx")
    if grad_zq is not None:
        F_hip._check_dev("grad_zq", grad_zq)
        grad_zq = grad_zq.contiguous()
        if grad_zq.shape != z_e.shape:
            raise ValueError("grad_zq must have z_e's shape")
    if grad_loss is not None:
        F_hip._check_dev("grad_loss", grad_loss)
        grad_loss = grad_loss.reshape(1).contiguous()
    dev = z_e.device
    L = _lib.load()
    with torch.cuda.device(dev):
        gz = torch.empty_like(z_e) if need_z else None
        ge = torch.empty_like(codebook) if need_codebook else None
        ws = None
        if need_codebook:
            n = L.vqvae_vq_backward_workspace_bytes(B * H * W, K, D)
            if n == 0:
                raise _lib.VqvaeHipError(f"VQ backward: shape N={B * H * W}, K={K}, D={D} not supported")
            ws = torch.empty(n, dtype=torch.uint8, device=dev)
        _lib.check(L.vqvae_vq_backward_f32(
            z_e.data_ptr(), codebook.data_ptr(), idx.data_ptr(),
            grad_zq.data_ptr() if grad_zq is not None else None,
            grad_loss.data_ptr() if grad_loss is not None else None,
            B, D, H, W, K, float(beta), F_hip.VQ_ROWMAJOR if rowmajor else 0,
            gz.data_ptr() if gz is not None else None, ge.data_ptr() if ge is not None else None,
            ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _sp(z_e)))
    return gz, ge


class VQStraightThrough(torch.autograd.Function):
    """(z_e, codebook) -> (loss, z_q, perplexity, idx, hist) with the reference's gradient structure
    [MEASURED in SURVEY.md 8b]: loss and z_q differentiable, perplexity / idx / hist not; d z_q / d z = I."""

    @staticmethod
    def forward(ctx, z_e, codebook, beta, rowmajor, workspace, prepared):
        z = z_e.detach().contiguous()
        w = codebook.detach().contiguous()
        loss, z_q, perplexity, idx, hist = F_hip.vq_forward(z, w, beta, rowmajor=rowmajor, workspace=workspace,
                                                            prepared=prepared)
        ctx.save_for_backward(z, w, idx)
        ctx.beta, ctx.rowmajor = beta, rowmajor
        ctx.mark_non_differentiable(perplexity, idx, hist)
        return loss, z_q, perplexity, idx, hist

    @staticmethod
    def backward(ctx, g_loss, g_zq, *_unused):
        z, w, idx = ctx.saved_tensors
        need_z, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if g_loss is None:
            g_loss = torch.zeros((), dtype=torch.float32, device=z.device)
        gz, gw = vq_backward(z, w, idx, g_zq, g_loss, ctx.beta, rowmajor=ctx.rowmajor, need_z=need_z,
                             need_codebook=need_w)
        return gz, gw, None, None, None, None


class _StepLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_hat, x, embedding_loss, perplexity, inv_var):
        F_hip._check_dev("x_hat", x_hat)
        F_hip._check_dev("x", x)
        xh, xx = x_hat.detach().contiguous(), x.detach().contiguous()
        if xh.shape != xx.shape:
            raise ValueError("x_hat and x must have the same shape")
        dev = xh.device
        L = _lib.load()
        with torch.cuda.device(dev):
            out = torch.empty(3, dtype=torch.float32, device=dev)
            ws = torch.empty(L.vqvae_recon_loss_workspace_bytes(), dtype=torch.uint8, device=dev)
            el = embedding_loss.detach().reshape(1).contiguous() if embedding_loss is not None else None
            pp = perplexity.detach().reshape(1).contiguous() if perplexity is not None else None
            _lib.check(L.vqvae_recon_loss_f32(xh.data_ptr(), xx.data_ptr(), xh.numel(), float(inv_var),
                                              el.data_ptr() if el is not None else None,
                                              pp.data_ptr() if pp is not None else None,
                                              out.data_ptr(), ws.data_ptr(), ws.numel(), _sp(xh)))
        ctx.save_for_backward(xh, xx)
        ctx.inv_var = float(inv_var)
        return out

    @staticmethod
    def backward(ctx, g):
        xh, xx = ctx.saved_tensors
        gx = gel = None
        if ctx.needs_input_grad[0]:
            gsum = (g[0] + g[1]).reshape(1).contiguous()       # recon_loss feeds out[0] and out[1]
            gx = torch.empty_like(xh)
            with torch.cuda.device(xh.device):
                _lib.check(_lib.load().vqvae_recon_loss_backward_f32(xh.data_ptr(), xx.data_ptr(), xh.numel(),
                                                                     ctx.inv_var, gsum.data_ptr(), gx.data_ptr(),
                                                                     _sp(xh)))
        if ctx.needs_input_grad[2]:
            gel = g[1].reshape(())
        return gx, None, gel, None, None


def step_losses(embedding_loss, x_hat, perplexity, x, x_train_var):
    """-> 3-element fp32 device tensor [recon_loss, loss, perplexity] (main.py:75-76, 81-83).

    `stats[1].backward()` is `loss.backward()` of main.py:78; `stats.tolist()` is the single D2H copy
    that replaces the three `.cpu()` calls of main.py:81-83."""
    return _StepLosses.apply(x_hat, x, embedding_loss, perplexity, 1.0 / float(x_train_var))
