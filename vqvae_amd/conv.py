"""Encoder / ResidualStack / Decoder forward drivers.

Two explicit backends, chosen by the caller and never switched silently:

  "hip"   (default) every conv, conv-transpose and residual layer is a hand-written
          gfx950 kernel from libvqvae_hip.so; activations stay row-major (B,H,W,C)
          between layers.  This is the product path (BASELINE config 3).
  "torch" the parameter-holding nn.Conv2d / nn.ConvTranspose2d modules are called
          as they are on the ROCm device ("torch convs unchanged", BASELINE config 2:
          only the VectorQuantizer is ours).  Opt-in via `set_conv_backend("torch")`;
          exists so config 2 can be measured, not as a fallback.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

_BACKEND = "hip"


def set_conv_backend(name: str):
    global _BACKEND
    if name not in ("hip", "torch"):
        raise ValueError(name)
    _BACKEND = name


def get_conv_backend() -> str:
    return _BACKEND


# --------------------------------------------------------------------------- torch backend
def _torch_residual(t, w_pairs, final_relu, relu_in=True):
    for w1, w2 in w_pairs:
        t = torch.relu(t)                       # the in-place ReLU also rewrites the skip
        t = t + F.conv2d(torch.relu(F.conv2d(t, w1, None, 1, 1)), w2)
    return torch.relu(t) if final_relu else t


def _torch_encoder(enc, x, pre_quant):
    cs = enc.conv_stack
    t = torch.relu(cs[0](x))
    t = torch.relu(cs[2](t))
    t = cs[4](t)
    t = _torch_residual(t, [(l.res_block[1].weight, l.res_block[3].weight) for l in cs[5].stack], True)
    if pre_quant is not None:
        t = pre_quant(t).permute(0, 2, 3, 1).contiguous()      # row-major for the VQ kernel
    return t


def _torch_decoder(dec, z_q, rowmajor_in):
    ds = dec.inverse_conv_stack
    if rowmajor_in:
        z_q = z_q.permute(0, 3, 1, 2)
    t = ds[0](z_q)
    t = _torch_residual(t, [(l.res_block[1].weight, l.res_block[3].weight) for l in ds[1].stack], True)
    t = torch.relu(ds[2](t))
    return ds[4](t)


# --------------------------------------------------------------------------- dispatch
def residual_stack_nchw(x, layers, final_relu):
    """ResidualLayer / ResidualStack forward at the NCHW module boundary (models/residual.py:27-29,47-51)."""
    if _BACKEND == "torch":
        return _torch_residual(x, [(l.res_block[1].weight, l.res_block[3].weight) for l in layers], final_relu)
    from . import conv_hip
    return conv_hip.residual_stack_nchw(x, layers, final_relu)


def encoder_forward(enc, x, pre_quant):
    """Encoder.forward (models/encoder.py:42-43) [+ pre_quantization_conv, models/vqvae.py:33].
    Returns NCHW when pre_quant is None (module boundary), else row-major (B,H/4,W/4,D)."""
    if _BACKEND == "torch":
        return _torch_encoder(enc, x, pre_quant)
    from . import conv_hip
    return conv_hip.encoder_forward(enc, x, pre_quant)


def decoder_forward(dec, z_q, rowmajor_in):
    """Decoder.forward (models/decoder.py:38-39); z_q NCHW, or row-major from the VQ kernel."""
    if _BACKEND == "torch":
        return _torch_decoder(dec, z_q, rowmajor_in)
    from . import conv_hip
    return conv_hip.decoder_forward(dec, z_q, rowmajor_in)
