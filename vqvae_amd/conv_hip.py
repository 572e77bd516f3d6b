"""HIP backend of vqvae_amd.conv: drives the conv / residual kernels of libvqvae_hip.so.

Activations stay row-major (B,H,W,C) between layers; packed weight images are cached on
the parameter-holding nn.Module (in a weak side table, vqvae_amd/_cache.py: never in the module's __dict__, so the
modules pickle and deep-copy like the reference's) and rebuilt when the parameter changes (data_ptr/_version).
"""
from __future__ import annotations

import torch

from . import _cache, _lib

CONV_4x4_S2, CONV_3x3_S1, CONV_1x1, CONVT_3x3_S1, CONVT_4x4_S2 = range(5)
RELU_IN, RELU_OUT = 1, 2
EXACT_FP32 = 4                 # VQVAE_CONV_EXACT_FP32


def _sp(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev_f32(name, t):
    if not t.is_cuda:
        raise _lib.VqvaeHipError(f"{name} must be on the GPU: the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the reference path is fp32 only)")


def _packed(mod, kind, weight, nbytes_fn, pack_fn):
    """Packed-weight cache on the owning module, keyed by the parameter's identity/version."""
    w = weight.detach()
    key = (kind, w.data_ptr(), w._version, str(w.device), tuple(w.shape))
    cache = _cache.side(mod).setdefault("packed", {})
    hit = cache.get(kind)
    if hit is not None and hit[0] == key:
        if hit[2] is not None:                       # packed on some stream a moment ago
            cur = torch.cuda.current_stream(w.device)
            if cur.cuda_stream == hit[2][1]:
                pass                                 # the packing stream itself: ordered behind the pack kernels already
            elif torch.cuda.is_current_stream_capturing():
                # neither an event query nor a wait on an event from outside the capture belongs in a capture.  Nothing to
                # do: a capture is preceded by an eager warm-up on the capture stream (GraphedForward; torch's own rule for its
                # allocator), which met this entry in the branch below and ordered the stream behind the pack kernels
                pass
            elif hit[2][0].query():                  # forget the event once it has completed (no per-call cost from then on)
                cache[kind] = (hit[0], hit[1], None)
            else:
                cur.wait_event(hit[2][0])
        return hit[1]
    n = nbytes_fn()
    if n == 0:
        raise _lib.VqvaeHipError(f"layer shape {tuple(w.shape)} not supported by the gfx950 conv kernels")
    buf = torch.empty(n // 4, dtype=torch.float32, device=w.device)
    _lib.check(pack_fn(w.contiguous(), buf))
    cache[kind] = (key, buf, _cache.mark_ready(w.device))
    return buf


def invalidate(mod):
    """Forget the packed-weight images cached on `mod` (see modules.VQVAE.invalidate_caches)."""
    _cache.drop(mod, "packed")


def _pack_conv(mod, kind, weight, Cin, Cout):
    L = _lib.load()
    return _packed(mod, ("conv", kind), weight,
                   lambda: L.vqvae_conv_packed_bytes(kind, Cin, Cout),
                   lambda w, buf: L.vqvae_conv_pack_f32(kind, w.data_ptr(), Cin, Cout, buf.data_ptr(), _sp(w)))


def conv(kind, x, mod, weight, bias, Cin, Cout, flags, addend=None, mask=None):
    """One nn.Conv2d / nn.ConvTranspose2d on row-major activations.  addend / mask (training's data-gradient launches):
    y = (mask > 0) ? conv + addend : 0 in the kernel's epilogue; where the kernel has no such epilogue the two passes run
    separately (torch add, vqvae_relu_backward_f32)."""
    B, H, W, C = x.shape
    assert C == Cin
    packed = _pack_conv(mod, kind, weight, Cin, Cout)
    if kind == CONV_4x4_S2:
        Ho, Wo = H // 2, W // 2
    elif kind == CONVT_4x4_S2:
        Ho, Wo = 2 * H, 2 * W
    else:
        Ho, Wo = H, W
    y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    b = bias.detach() if bias is not None else None
    L = _lib.load()
    if addend is not None or mask is not None:
        for t in (addend, mask):
            if t is not None and (t.shape != y.shape or not t.is_contiguous()):
                raise ValueError("addend / mask must be contiguous with the output's shape")
        rc = L.vqvae_conv_forward_ep_f32(kind, x.data_ptr(), packed.data_ptr(), b.data_ptr() if b is not None else None,
                                         B, H, W, Cin, Cout, flags, addend.data_ptr() if addend is not None else None,
                                         mask.data_ptr() if mask is not None else None, y.data_ptr(), _sp(x))
        if rc == 0:
            return y
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc)
    _lib.check(L.vqvae_conv_forward_f32(kind, x.data_ptr(), packed.data_ptr(), b.data_ptr() if b is not None else None,
                                        B, H, W, Cin, Cout, flags, y.data_ptr(), _sp(x)))
    if addend is not None:
        y = y + addend
    if mask is not None:
        out = torch.empty_like(y)
        _lib.check(L.vqvae_relu_backward_f32(y.data_ptr(), mask.data_ptr(), y.numel(), out.data_ptr(), _sp(y)))
        y = out
    return y


def conv_taps(x, mod, weight, bias, taps, flags=0):
    """A stride-1 convolution over an explicit tap list [(dy, dx), ...] on row-major activations: weight (Cout, Cin, kh, kw) with
    kh * kw == len(taps), the list in the weight's (ky, kx) order -- a masked convolution without an im2col pass
    (vqvae_conv_taps_forward_f32).  Lists of more than 16 taps run as a chain of launches over slices of at most 16, each adding to
    the previous one's result in its epilogue (the sum over taps in slice order instead of one accumulator chain)."""
    import ctypes as C
    B, H, W, Cin = x.shape
    Cout, n = weight.shape[0], len(taps)
    if weight.shape[1] != Cin or weight.shape[2] * weight.shape[3] != n:
        raise ValueError("weight must be (Cout, Cin, kh, kw) with kh * kw taps")
    L = _lib.load()
    b = bias.detach() if bias is not None else None
    nsl = (n + 15) // 16
    if nsl > 1 and (flags & RELU_OUT):
        raise ValueError("an output ReLU cannot ride on a chain of tap slices")
    per = (n + nsl - 1) // nsl
    y = None
    for s0 in range(0, n, per):
        sl = taps[s0:s0 + per]
        k = len(sl)
        dy = (C.c_int8 * k)(*[t[0] for t in sl])
        dx = (C.c_int8 * k)(*[t[1] for t in sl])
        key = ("taps", tuple(sl), s0)
        def pack(w, buf, dy=dy, dx=dx, k=k, s0=s0):
            src = w if nsl == 1 else w.reshape(Cout, Cin, n)[:, :, s0:s0 + k].contiguous()
            rc = L.vqvae_conv_taps_pack_f32(src.data_ptr(), k, C.cast(dy, C.c_void_p), C.cast(dx, C.c_void_p), Cin, Cout, buf.data_ptr(), _sp(w))
            return rc                  # (a slice's temporary is freed in stream order behind the pack launch: same stream)
        packed = _packed(mod, key, weight, lambda k=k: L.vqvae_conv_taps_packed_bytes(k, Cin, Cout), pack)
        out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
        _lib.check(L.vqvae_conv_taps_forward_ep_f32(x.data_ptr(), packed.data_ptr(), b.data_ptr() if (b is not None and y is None) else None,
                                                    B, H, W, Cin, Cout, k, C.cast(dy, C.c_void_p), C.cast(dx, C.c_void_p), flags,
                                                    y.data_ptr() if y is not None else None, None, out.data_ptr(), _sp(x)))
        y = out
    return y


def res_layer(x, layer, flags, want_hidden=False):
    """One ResidualLayer on row-major activations (fused kernel).  want_hidden: also return relu(W1 * r(x)) as
    (B,H,W,Rh) where the kernel can write it (8x8 maps, res_h = 32), else None -- a training step keeps it for backward."""
    c1, c2 = layer.res_block[1], layer.res_block[3]
    B, H, W, C = x.shape
    Rh = c1.weight.shape[0]
    if C % 4 or Rh % 4:
        raise _lib.VqvaeHipError(f"residual layer C={C}, res_h={Rh}: the gfx950 kernels need multiples of 4 channels")
    p1 = _pack_conv(c1, CONV_3x3_S1, c1.weight, C, Rh)
    p2 = _pack_conv(c2, CONV_1x1, c2.weight, Rh, C)
    y = torch.empty_like(x)
    if Rh > 32 or C not in (32, 64, 128):
        # widths outside the fused kernel (round 4): 3x3 conv -> 1x1 conv -> skip + ReLU through the conv kernels
        scratch = torch.empty((B, H, W, Rh), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().vqvae_res_layer_forward_ws_f32(x.data_ptr(), p1.data_ptr(), p2.data_ptr(), B, H, W, C, Rh, flags,
                                                              y.data_ptr(), scratch.data_ptr(), scratch.numel() * 4, _sp(x)))
        return (y, None) if want_hidden else y
    if want_hidden and H == 8 and W == 8 and Rh == 32:
        hid = torch.empty((B, H, W, Rh), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().vqvae_res_layer_forward_hidden_f32(x.data_ptr(), p1.data_ptr(), p2.data_ptr(), B, H, W, C, Rh,
                                                                  flags, y.data_ptr(), hid.data_ptr(), _sp(x)))
        return y, hid
    _lib.check(_lib.load().vqvae_res_layer_forward_f32(x.data_ptr(), p1.data_ptr(), p2.data_ptr(), B, H, W, C, Rh,
                                                       flags, y.data_ptr(), _sp(x)))
    return (y, None) if want_hidden else y


def transpose(x, batch, R, Cc):
    y = torch.empty_like(x)
    _lib.check(_lib.load().vqvae_transpose_f32(x.data_ptr(), batch, R, Cc, y.data_ptr(), _sp(x)))
    return y


def nchw_to_rows(x):
    B, C, H, W = x.shape
    return transpose(x.contiguous(), B, C, H * W).view(B, H, W, C)


def rows_to_nchw(x):
    B, H, W, C = x.shape
    return transpose(x.contiguous(), B, H * W, C).view(B, C, H, W)


def _res_stack_rows(t, layers, first_relu_in, final_relu):
    """t row-major.  Every layer's output feeds either the next layer's in-place ReLU or the
    stack's final ReLU, so the ReLU is applied once, by the producer."""
    n = len(layers)
    for i, layer in enumerate(layers):
        flags = (RELU_IN if (i == 0 and first_relu_in) else 0)
        if i < n - 1 or final_relu:
            flags |= RELU_OUT
        t = res_layer(t, layer, flags)
    if n == 0 and final_relu:
        t = torch.relu(t)
    return t


def residual_stack_nchw(x, layers, final_relu):
    _dev_f32("x", x)
    t = _res_stack_rows(nchw_to_rows(x), layers, True, final_relu)
    return rows_to_nchw(t)


def encoder_forward(enc, x, pre_quant):
    """models/encoder.py:28-43 (+ models/vqvae.py:33 when pre_quant is given)."""
    _dev_f32("x", x)
    cs = enc.conv_stack
    c0, c2, c4, stack = cs[0], cs[2], cs[4], cs[5]
    x = x.contiguous()
    B, Cin, H, W = x.shape
    if Cin != c0.weight.shape[1]:
        raise ValueError(f"expected {c0.weight.shape[1]} input channels, got {Cin}")
    L = _lib.load()
    C0 = c0.weight.shape[0]
    p0 = _packed(c0, ("conv_in",), c0.weight,
                 lambda: L.vqvae_conv_in_packed_bytes(Cin, C0),
                 lambda w, buf: L.vqvae_conv_in_pack_f32(w.data_ptr(), Cin, C0, buf.data_ptr(), _sp(w)))
    a0 = torch.empty((B, H // 2, W // 2, C0), dtype=torch.float32, device=x.device)
    _lib.check(L.vqvae_conv_in_forward_f32(x.data_ptr(), p0.data_ptr(), c0.bias.detach().data_ptr(), B, H, W, Cin,
                                           C0, RELU_OUT, a0.data_ptr(), _sp(x)))                       # :29-31
    C1 = c2.weight.shape[0]
    a1 = conv(CONV_4x4_S2, a0, c2, c2.weight, c2.bias, C0, C1, RELU_OUT)                               # :32-34
    # conv_stack[4] feeds ResidualStack, whose first in-place ReLU (residual.py:19) or final
    # F.relu (:50) is the only consumer -> fuse that ReLU here
    a2 = conv(CONV_3x3_S1, a1, c4, c4.weight, c4.bias, C1, c4.weight.shape[0], RELU_OUT)               # :35-36
    t = _res_stack_rows(a2, list(stack.stack), False, True)                                            # :37-38
    if pre_quant is None:
        return rows_to_nchw(t)
    D = pre_quant.weight.shape[0]
    return conv(CONV_1x1, t, pre_quant, pre_quant.weight, pre_quant.bias, t.shape[3], D, 0)            # vqvae.py:33


def decoder_forward(dec, z_q, rowmajor_in):
    """models/decoder.py:27-39."""
    _dev_f32("z_q", z_q)
    ds = dec.inverse_conv_stack
    d0, stack, d2, d4 = ds[0], ds[1], ds[2], ds[4]
    t = z_q.contiguous() if rowmajor_in else nchw_to_rows(z_q)
    Din = d0.weight.shape[0]
    if t.shape[3] != Din:
        raise ValueError(f"expected {Din} latent channels, got {t.shape[3]}")
    h_dim = d0.weight.shape[1]
    a0 = conv(CONVT_3x3_S1, t, d0, d0.weight, d0.bias, Din, h_dim, RELU_OUT)                           # :28-29 (+ stack ReLU)
    a1 = _res_stack_rows(a0, list(stack.stack), False, True)                                           # :30
    C2 = d2.weight.shape[1]
    a2 = conv(CONVT_4x4_S2, a1, d2, d2.weight, d2.bias, h_dim, C2, RELU_OUT)                           # :31-33
    B, H2, W2, _ = a2.shape
    Cout = d4.weight.shape[1]
    L = _lib.load()
    p4 = _packed(d4, ("convt_out",), d4.weight,
                 lambda: L.vqvae_convt_out_packed_bytes(C2, Cout),
                 lambda w, buf: L.vqvae_convt_out_pack_f32(w.data_ptr(), C2, Cout, buf.data_ptr(), _sp(w)))
    x_hat = torch.empty((B, Cout, 2 * H2, 2 * W2), dtype=torch.float32, device=a2.device)
    _lib.check(L.vqvae_convt_out_forward_f32(a2.data_ptr(), p4.data_ptr(), d4.bias.detach().data_ptr(), B, H2, W2,
                                             C2, Cout, 0, x_hat.data_ptr(), _sp(a2)))                     # :34-35
    return x_hat
