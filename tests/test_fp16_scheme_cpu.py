"""CPU emulation of the two-term fp16 product scheme's OPERAND SCALES (no GPU): why round 4 moved the weight scale from one
power of two per tensor to one per output channel.

x = h1 + h2 + r (fp16 terms of x * 2^k), products h1 g1 + h1 g2 + h2 g1 -- emulated exactly in fp64 (the accumulation error
is not part of this question) for the encoder's 3x3 128 -> 128 layer on trained-checkpoint-like weights (tests/hetero.py):
  * per-TENSOR weight scale: a row 2^17 below the tensor's maximum loses the bits fp32 keeps -- on the diagonal
    re-parametrisation ("coupled") the error reaches TENS OF PER CENT of an output channel's maximum;
  * per-OUTPUT-CHANNEL weight scale (what conv_wscale_kernel packs now): the error stays at the fp32 convolution's own level,
    a few 1e-7 of the channel maximum.
The GPU counterpart is tests/test_parity_hetero_gpu.py."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import torch_port
from tests import hetero

E = "encoder.conv_stack."


def _split(x, scale):
    xs = (x * scale).float()
    h1 = xs.half()
    h2 = (xs - h1.float()).half()
    return h1.double() / scale, h2.double() / scale


def _p2(m):
    return 2.0 ** (14 - torch.floor(torch.log2(m.clamp_min(1e-300))))


def _errors(sd):
    x = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        t = F.relu(F.conv2d(x, sd[E + "0.weight"], sd[E + "0.bias"], 2, 1))
        t = F.relu(F.conv2d(t, sd[E + "2.weight"], sd[E + "2.bias"], 2, 1)).double()
        w = sd[E + "4.weight"].double()
        exact = F.conv2d(t, w, None, 1, 1)
        cmax = exact.abs().amax(dim=(0, 2, 3), keepdim=True)
        fp32 = float(((F.conv2d(t.float(), w.float(), None, 1, 1).double() - exact).abs() / cmax).max())
        x1, x2 = _split(t, _p2(t.abs().amax(dim=(1, 2, 3), keepdim=True)))          # one scale per image
        out = {}
        for mode in ("tensor", "channel"):
            sw = _p2(w.abs().max()) if mode == "tensor" else _p2(w.abs().amax(dim=(1, 2, 3), keepdim=True))
            w1, w2 = _split(w, sw)
            y = F.conv2d(x1, w1, None, 1, 1) + F.conv2d(x1, w2, None, 1, 1) + F.conv2d(x2, w1, None, 1, 1)
            out[mode] = float(((y - exact).abs() / cmax).max())
    return out, fp32


def test_per_channel_weight_scale_is_what_keeps_the_fp16_scheme_fp32_grade():
    sd0 = torch_port.init_state_dict()
    # default init: one magnitude per layer by construction -- both scale choices are the same thing
    e, f32 = _errors(sd0)
    assert e["tensor"] == e["channel"] and e["channel"] < 1e-6
    # trained-like, coupled: the per-tensor scale is catastrophic, the per-channel one fp32-grade
    e, f32 = _errors(hetero.rescale_coupled(sd0, 1))
    assert e["tensor"] > 1e-3, e
    assert e["channel"] < 4e-6 and e["channel"] < 10 * f32, (e, f32)
    # trained-like, independent factors: per tensor already outside the 1e-5 tolerance, per channel inside
    e, f32 = _errors(hetero.rescale_independent(sd0, 1))
    assert e["tensor"] > 5e-6 and e["channel"] < 2e-6, (e, f32)


def test_coupled_rescaling_is_a_reparametrisation():
    """tests/hetero.rescale_coupled keeps the function (fp64): z_e / x_hat are the default model's up to the channel factors."""
    sd0 = torch_port.init_state_dict()
    sd = hetero.rescale_coupled(sd0, 3, codebook=False)
    x = torch.randn(4, 3, 32, 32, generator=torch.Generator().manual_seed(2)).double()
    d0 = {k: v.double() for k, v in sd0.items()}
    d1 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        z0, z1 = torch_port.encode(d0, x.clone(), 2), torch_port.encode(d1, x.clone(), 2)
    ratio = (z1.abs().amax(dim=(0, 2, 3)) / z0.abs().amax(dim=(0, 2, 3)))
    np.testing.assert_allclose((z1 / ratio.view(1, -1, 1, 1)).numpy(), z0.numpy(), rtol=1e-4, atol=1e-7)
    assert float(ratio.max() / ratio.min()) > 1e3
