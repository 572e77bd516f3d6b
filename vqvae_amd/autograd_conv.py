"""Autograd for the HIP conv path: `loss.backward()` of main.py:78 with every conv,
conv-transpose and residual layer on libvqvae_hip.so, forward AND backward.

  data gradients    the forward kernels with the mirrored `kind` and the SAME weight tensor
                    (d/dx Conv2d = ConvTranspose2d and vice versa; last layer: the first-layer kernel)
  weight gradients  vqvae_conv_wgrad_ex_f32 (two-term fp16 products on the 8x8-map layers, exact fp32 MFMA elsewhere or with
                    WGRAD_EXACT_FP32; fixed-order reduction, bit-reproducible)
  bias gradients    vqvae_bias_grad_f32;   ReLU masks  vqvae_relu_backward_f32

ReLU masks and skip sums ride in the data-gradient kernels' epilogues (round 4, vqvae_conv_forward_ep_f32): a layer whose INPUT
is the ReLU'd output of the layer below (`x_is_relu`) returns its input gradient already multiplied by (x > 0) -- that IS the
ReLU backward of the layer below, which is told `consumer_masks` and skips its own pass; a residual layer's skip gradient is
the epilogue's addend.  FUSE_EPILOGUES = False restores the separate passes (tests compare the two).

Activations are row-major (B,H,W,C) between layers, as in the forward-only path.  ResidualLayer keeps its
fused forward kernel, which also writes the hidden activation for backward on 8x8 maps (other shapes: recomputed in
backward).  No CPU path, no fallback.
"""
from __future__ import annotations

import torch

from . import _cache, _lib, conv_hip
from .conv_hip import (CONV_1x1, CONV_3x3_S1, CONV_4x4_S2, CONVT_3x3_S1, CONVT_4x4_S2, RELU_IN, RELU_OUT, _sp)

CONVT_1x1 = 5
# kind -> (kernel size, stride, pad, is_transposed, kind of the data-gradient conv)
_GEOM = {
    CONV_4x4_S2: (4, 2, 1, False, CONVT_4x4_S2),
    CONV_3x3_S1: (3, 1, 1, False, CONVT_3x3_S1),
    CONV_1x1: (1, 1, 0, False, CONVT_1x1),
    CONVT_3x3_S1: (3, 1, 1, True, CONV_3x3_S1),
    CONVT_4x4_S2: (4, 2, 1, True, CONV_4x4_S2),
}


FUSE_EPILOGUES = True
WGRAD_EXACT_FP32 = False        # True: every weight gradient on the exact-fp32 MFMA kernels (round 3's arithmetic); default: two-term
                                # fp16 products on the map-resident shapes (vqvae_conv_wgrad_ex_f32)


class _Holder:
    """Stand-in 'module' that owns the packed-weight cache of a data-gradient conv."""


def _holder(mod, tag):
    h = _cache.side(mod).setdefault("bwd", {})
    if tag not in h:
        h[tag] = _Holder()
    return h[tag]


def relu_backward(g, y):
    g = g.contiguous()
    out = torch.empty_like(g)
    _lib.check(_lib.load().vqvae_relu_backward_f32(g.data_ptr(), y.data_ptr(), g.numel(), out.data_ptr(), _sp(g)))
    return out


def bias_grad(g, nchw=False):
    g = g.contiguous()
    if nchw:
        B, C, H, W = g.shape
    else:
        B, H, W, C = g.shape
    L = _lib.load()
    ws = torch.empty(L.vqvae_bias_grad_workspace_bytes(C), dtype=torch.uint8, device=g.device)
    out = torch.empty(C, dtype=torch.float32, device=g.device)
    _lib.check(L.vqvae_bias_grad_f32(g.data_ptr(), B, H * W, C, 1 if nchw else 0, out.data_ptr(), ws.data_ptr(),
                                     ws.numel(), _sp(g)))
    return out


def conv_wgrad(a_rows, bt, k, stride, pad, bt_nchw=False):
    """grad_w[ca][cb][ky][kx] = sum a[b,y,x,ca] * bt[b, y*s+ky-p, x*s+kx-p, cb]  (see include/vqvae_hip.h)."""
    a_rows = a_rows.contiguous()
    bt = bt.contiguous()
    B, HA, WA, CA = a_rows.shape
    if bt_nchw:
        _, CB, HB, WB = bt.shape
    else:
        _, HB, WB, CB = bt.shape
    L = _lib.load()
    n = L.vqvae_conv_wgrad_workspace_bytes(CA, CB, k)
    ws = torch.empty(n, dtype=torch.uint8, device=a_rows.device)
    gw = torch.empty((CA, CB, k, k), dtype=torch.float32, device=a_rows.device)
    _lib.check(L.vqvae_conv_wgrad_ex_f32(a_rows.data_ptr(), bt.data_ptr(), B, HA, WA, CA, HB, WB, CB, k, stride, pad,
                                         1 if bt_nchw else 0, conv_hip.EXACT_FP32 if WGRAD_EXACT_FP32 else 0, gw.data_ptr(),
                                         ws.data_ptr(), n, _sp(a_rows)))
    return gw


class ConvFn(torch.autograd.Function):
    """One nn.Conv2d / nn.ConvTranspose2d (+ optional fused output ReLU) on row-major activations."""

    @staticmethod
    def forward(ctx, x, weight, bias, mod, kind, relu_out, x_is_relu=False, consumer_masks=False):
        k, s, p, transposed, _ = _GEOM[kind]
        Cin, Cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
        y = conv_hip.conv(kind, x.detach().contiguous(), mod, weight.detach(), bias.detach() if bias is not None else None,
                          Cin, Cout, RELU_OUT if relu_out else 0)
        own_mask = relu_out and not consumer_masks           # the ReLU backward of this layer's output runs here
        ctx.save_for_backward(x.detach().contiguous(), weight.detach(), y if own_mask else None)
        ctx.mod, ctx.kind, ctx.own_mask, ctx.has_bias, ctx.x_is_relu = mod, kind, own_mask, bias is not None, x_is_relu
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        k, s, p, transposed, dkind = _GEOM[ctx.kind]
        Cin, Cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        gy = relu_backward(gy, y) if ctx.own_mask else gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # data gradient: the mirrored conv, same weight tensor, channels swapped (+ the ReLU mask of the layer below)
            gx = conv_hip.conv(dkind, gy, _holder(ctx.mod, ("dgrad", ctx.kind)), w, None, Cout, Cin, 0,
                               mask=x if ctx.x_is_relu else None)
        if ctx.needs_input_grad[1]:
            gw = conv_wgrad(x, gy, k, s, p) if transposed else conv_wgrad(gy, x, k, s, p)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = bias_grad(gy)
        return gx, gw, gb, None, None, None, None, None


class ConvInFn(torch.autograd.Function):
    """First layer: nn.Conv2d(Cin<=4, C0, 4, 2, 1) + ReLU on the NCHW image (models/encoder.py:29-31)."""

    @staticmethod
    def forward(ctx, x_nchw, weight, bias, mod, consumer_masks=False):
        L = _lib.load()
        x = x_nchw.detach().contiguous()
        B, Cin, H, W = x.shape
        C0 = weight.shape[0]
        p0 = conv_hip._packed(mod, ("conv_in",), weight, lambda: L.vqvae_conv_in_packed_bytes(Cin, C0),
                              lambda w, buf: L.vqvae_conv_in_pack_f32(w.data_ptr(), Cin, C0, buf.data_ptr(), _sp(w)))
        y = torch.empty((B, H // 2, W // 2, C0), dtype=torch.float32, device=x.device)
        _lib.check(L.vqvae_conv_in_forward_f32(x.data_ptr(), p0.data_ptr(), bias.detach().data_ptr(), B, H, W, Cin, C0,
                                               RELU_OUT, y.data_ptr(), _sp(x)))
        ctx.save_for_backward(x, None if consumer_masks else y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y = ctx.saved_tensors
        gy = relu_backward(gy, y) if y is not None else gy.contiguous()
        gw = conv_wgrad(gy, x, 4, 2, 1, bt_nchw=True) if ctx.needs_input_grad[1] else None
        gb = bias_grad(gy) if ctx.needs_input_grad[2] else None
        return None, gw, gb, None, None


class ConvTOutFn(torch.autograd.Function):
    """Last layer: nn.ConvTranspose2d(C, Cout<=4, 4, 2, 1), row-major in, NCHW image out (decoder.py:34-35)."""

    @staticmethod
    def forward(ctx, t, weight, bias, mod, x_is_relu=False):
        L = _lib.load()
        t = t.detach().contiguous()
        B, H, W, C = t.shape
        Cout = weight.shape[1]
        p4 = conv_hip._packed(mod, ("convt_out",), weight, lambda: L.vqvae_convt_out_packed_bytes(C, Cout),
                              lambda w, buf: L.vqvae_convt_out_pack_f32(w.data_ptr(), C, Cout, buf.data_ptr(), _sp(w)))
        x_hat = torch.empty((B, Cout, 2 * H, 2 * W), dtype=torch.float32, device=t.device)
        _lib.check(L.vqvae_convt_out_forward_f32(t.data_ptr(), p4.data_ptr(), bias.detach().data_ptr(), B, H, W, C, Cout, 0,
                                                 x_hat.data_ptr(), _sp(t)))
        ctx.save_for_backward(t, weight.detach())
        ctx.mod, ctx.x_is_relu = mod, x_is_relu
        return x_hat

    @staticmethod
    def backward(ctx, g):
        t, w = ctx.saved_tensors
        g = g.contiguous()
        B, H, W, C = t.shape
        Cout = w.shape[1]
        L = _lib.load()
        gt = gw = gb = None
        if ctx.needs_input_grad[0]:
            # d/dt = Conv2d(Cout -> C, 4, 2, 1) of the NCHW gradient; the (C, Cout, 4, 4) weight tensor IS that
            # conv's (out, in, kh, kw) weight: the first-layer kernel does it
            hold = _holder(ctx.mod, ("dgrad", "convt_out"))
            pk = conv_hip._packed(hold, ("conv_in",), w, lambda: L.vqvae_conv_in_packed_bytes(Cout, C),
                                  lambda wt, buf: L.vqvae_conv_in_pack_f32(wt.data_ptr(), Cout, C, buf.data_ptr(), _sp(wt)))
            gt = torch.empty_like(t)
            rc = _lib.ERR_UNSUPPORTED
            if ctx.x_is_relu:                             # (t > 0) in the kernel's epilogue: the ReLU backward of the layer below
                rc = L.vqvae_conv_in_forward_ep_f32(g.data_ptr(), pk.data_ptr(), None, B, 2 * H, 2 * W, Cout, C, 0, t.data_ptr(),
                                                    gt.data_ptr(), _sp(g))
                if rc not in (0, _lib.ERR_UNSUPPORTED):
                    _lib.check(rc)
            if rc != 0:
                _lib.check(L.vqvae_conv_in_forward_f32(g.data_ptr(), pk.data_ptr(), None, B, 2 * H, 2 * W, Cout, C, 0,
                                                       gt.data_ptr(), _sp(g)))
                if ctx.x_is_relu:
                    gt = relu_backward(gt, t)
        if ctx.needs_input_grad[1]:
            gw = conv_wgrad(t, g, 4, 2, 1, bt_nchw=True)
        if ctx.needs_input_grad[2]:
            gb = bias_grad(g, nchw=True)
        return gt, gw, gb, None, None


class ResLayerFn(torch.autograd.Function):
    """One ResidualLayer (models/residual.py:18-29): fused forward kernel; it also writes the hidden activation for
    backward on 8x8 maps (recomputed in backward elsewhere).  y = [relu](r + W2 * relu(W1 * r)),  r = relu(x) if relu_in else x."""

    @staticmethod
    def forward(ctx, x, w1, w2, layer, relu_in, relu_out, x_is_relu=False, consumer_masks=False):
        flags = (RELU_IN if relu_in else 0) | (RELU_OUT if relu_out else 0)
        y, hid = conv_hip.res_layer(x.detach().contiguous(), layer, flags, want_hidden=True)
        own_mask = relu_out and not consumer_masks
        ctx.save_for_backward(x.detach().contiguous(), w1.detach(), w2.detach(), y if own_mask else None, hid)
        ctx.layer, ctx.relu_in, ctx.own_mask, ctx.x_is_relu = layer, relu_in, own_mask, x_is_relu
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w1, w2, y, h = ctx.saved_tensors
        c1, c2 = ctx.layer.res_block[1], ctx.layer.res_block[3]
        C, Rh = w1.shape[1], w1.shape[0]
        g = relu_backward(gy, y) if ctx.own_mask else gy.contiguous()
        if h is None:
            # h = relu(W1 * r) again with the plain conv kernel (shapes whose fused forward kernel does not write it)
            h = conv_hip.conv(CONV_3x3_S1, x, _holder(ctx.layer, "w1_fwd"), w1, None, C, Rh,
                              (RELU_IN if ctx.relu_in else 0) | RELU_OUT)
        # (B,H,W,Rh), times (h > 0): the hidden ReLU's backward in the 1x1 data gradient's epilogue
        gh = conv_hip.conv(CONVT_1x1, g, _holder(ctx.layer, "w2_dgrad"), w2, None, C, Rh, 0, mask=h)
        gx = gw1 = gw2 = None
        if ctx.needs_input_grad[0]:
            # skip + block: g + W1^T gh, times (x > 0) for the layer's own in-place ReLU and / or the ReLU of the layer below
            gx = conv_hip.conv(CONVT_3x3_S1, gh, _holder(ctx.layer, "w1_dgrad"), w1, None, Rh, C, 0, addend=g,
                               mask=x if (ctx.relu_in or ctx.x_is_relu) else None)
        r = torch.relu(x) if ctx.relu_in else x
        if ctx.needs_input_grad[1]:
            gw1 = conv_wgrad(gh, r, 3, 1, 1)                      # (Rh, C, 3, 3)
        if ctx.needs_input_grad[2]:
            gw2 = conv_wgrad(g, h, 1, 1, 0)                       # (C, Rh, 1, 1)
        return gx, gw1, gw2, None, None, None, None, None


def _res_stack_train(t, layers, first_relu_in, final_relu, x_is_relu=False, consumer_masks=False):
    """x_is_relu: t is the ReLU'd output of a layer that was told consumer_masks; consumer_masks: the consumer of the
    stack's (ReLU'd) output applies that ReLU's backward."""
    n = len(layers)
    for i, layer in enumerate(layers):
        relu_in = i == 0 and first_relu_in
        relu_out = i < n - 1 or final_relu
        # inside the stack every layer's output feeds exactly the next layer, which masks with its own input
        t = ResLayerFn.apply(t, layer.res_block[1].weight, layer.res_block[3].weight, layer, relu_in, relu_out,
                             (x_is_relu if i == 0 else FUSE_EPILOGUES), (consumer_masks if i == n - 1 else FUSE_EPILOGUES))
    if n == 0 and final_relu:
        t = torch.relu(t)
    return t


def encoder_forward_train(enc, x, pre_quant):
    """models/encoder.py:28-43 (+ models/vqvae.py:33) under autograd; returns row-major z_e."""
    cs = enc.conv_stack
    c0, c2, c4, stack = cs[0], cs[2], cs[4], cs[5]
    f = FUSE_EPILOGUES
    layers = list(stack.stack)
    fs = f and len(layers) > 0          # (no residual layers: torch.relu sits between the 3x3 conv and the 1x1 conv)
    a0 = ConvInFn.apply(x, c0.weight, c0.bias, c0, f)
    a1 = ConvFn.apply(a0, c2.weight, c2.bias, c2, CONV_4x4_S2, True, f, f)
    a2 = ConvFn.apply(a1, c4.weight, c4.bias, c4, CONV_3x3_S1, True, f, fs)      # + the stack's first in-place ReLU
    t = _res_stack_train(a2, layers, False, True, fs, fs)
    return ConvFn.apply(t, pre_quant.weight, pre_quant.bias, pre_quant, CONV_1x1, False, fs, False)


def encoder_stack_forward_train(enc, x):
    """Encoder.forward alone under autograd (models/encoder.py:42-43): NCHW image -> NCHW (B, h_dim, H/4, W/4), every layer
    forward and backward on the HIP kernels (round 5: sub-modules are differentiable like the reference's)."""
    cs = enc.conv_stack
    c0, c2, c4, stack = cs[0], cs[2], cs[4], cs[5]
    f = FUSE_EPILOGUES
    layers = list(stack.stack)
    fs = f and len(layers) > 0
    a0 = ConvInFn.apply(x, c0.weight, c0.bias, c0, f)
    a1 = ConvFn.apply(a0, c2.weight, c2.bias, c2, CONV_4x4_S2, True, f, f)
    a2 = ConvFn.apply(a1, c4.weight, c4.bias, c4, CONV_3x3_S1, True, f, fs)
    t = _res_stack_train(a2, layers, False, True, fs, False)         # (no consumer kernel behind the stack: it masks itself)
    return t.permute(0, 3, 1, 2).contiguous()


def residual_stack_forward_train(x_nchw, layers, final_relu):
    """ResidualStack / ResidualLayer.forward alone under autograd (models/residual.py:28, :47-51): NCHW in, NCHW out."""
    t = _res_stack_train(x_nchw.permute(0, 2, 3, 1).contiguous(), layers, True, final_relu, False, False)
    return t.permute(0, 3, 1, 2).contiguous()


def decoder_forward_train(dec, z_q_rows):
    """models/decoder.py:27-39 under autograd; z_q row-major (B,h,w,D) -> x_hat NCHW."""
    ds = dec.inverse_conv_stack
    d0, stack, d2, d4 = ds[0], ds[1], ds[2], ds[4]
    f = FUSE_EPILOGUES
    layers = list(stack.stack)
    fs = f and len(layers) > 0
    a0 = ConvFn.apply(z_q_rows, d0.weight, d0.bias, d0, CONVT_3x3_S1, True, False, fs)
    a1 = _res_stack_train(a0, layers, False, True, fs, fs)
    a2 = ConvFn.apply(a1, d2.weight, d2.bias, d2, CONVT_4x4_S2, True, fs, f)
    return ConvTOutFn.apply(a2, d4.weight, d4.bias, d4, f)
