"""GPU parity tests (-m gpu) for the conv / conv-transpose / residual kernels, through the C ABI.

Floating-point tolerance (stated per SURVEY.md 8c, tier P1): convs compute in exact fp32 on the
matrix cores but oneDNN's summation order is opaque, so every layer is compared with
    |y - y_ref| <= 1e-5 + 1e-4 * |y_ref|
against (a) torch's CPU fp32 op (what the reference executes) and (b) the C oracle (correctly
rounded, double accumulation).  Observed errors are ~1e-7.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ATOL, RTOL = 1e-5, 1e-4


def dev():
    return torch.device("cuda:0")


def rows(x):      # NCHW -> (B,H,W,C)
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def close(a, b):
    np.testing.assert_allclose(a, b, atol=ATOL, rtol=RTOL)


CONV_CASES = [
    # kind name, module ctor, B, Cin, H, W
    ("conv4x4s2", lambda ci, co: nn.Conv2d(ci, co, 4, 2, 1), 3, 64, 128, 16, 16),
    ("conv4x4s2", lambda ci, co: nn.Conv2d(ci, co, 4, 2, 1), 2, 32, 64, 12, 20),
    ("conv4x4s2", lambda ci, co: nn.Conv2d(ci, co, 4, 2, 1), 5, 32, 64, 16, 16),     # space-to-depth tile kernel, ragged batch
    ("conv3x3", lambda ci, co: nn.Conv2d(ci, co, 3, 1, 1), 3, 128, 128, 8, 8),
    ("conv3x3", lambda ci, co: nn.Conv2d(ci, co, 3, 1, 1, bias=False), 2, 128, 32, 7, 9),
    ("conv3x3", lambda ci, co: nn.Conv2d(ci, co, 3, 1, 1), 1, 16, 48, 5, 5),      # Cin < 32, Cout not /32
    ("conv1x1", lambda ci, co: nn.Conv2d(ci, co, 1, 1), 5, 128, 64, 8, 8),
    ("conv1x1", lambda ci, co: nn.Conv2d(ci, co, 1, 1, bias=False), 2, 32, 128, 8, 8),
    ("convT3x3", lambda ci, co: nn.ConvTranspose2d(ci, co, 3, 1, 1), 3, 64, 128, 8, 8),
    ("convT3x3", lambda ci, co: nn.ConvTranspose2d(ci, co, 3, 1, 1), 2, 128, 96, 6, 10),
    ("convT4x4s2", lambda ci, co: nn.ConvTranspose2d(ci, co, 4, 2, 1), 3, 128, 64, 8, 8),
    ("convT4x4s2", lambda ci, co: nn.ConvTranspose2d(ci, co, 4, 2, 1), 2, 64, 32, 5, 7),
]
KIND = {"conv4x4s2": 0, "conv3x3": 1, "conv1x1": 2, "convT3x3": 3, "convT4x4s2": 4}


@pytest.mark.parametrize("exact", [0, 4, 8], ids=["default", "fp32mfma", "bf16split"])
@pytest.mark.parametrize("relu_in,relu_out", [(False, False), (True, True)])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"{c[0]}-{c[3]}to{c[4]}-{c[5]}x{c[6]}")
def test_conv_layer_vs_torch_cpu(case, relu_in, relu_out, exact):
    """All product paths: the default (two-term fp16 products with per-image / per-layer power-of-two scales on 8x8 maps,
    three-term bf16 products elsewhere; fp32 accumulate), the three-term bf16 one everywhere (flag 0x8) and the
    exact-fp32 MFMA one (flag 0x4)."""
    from vqvae_amd import conv_hip
    name, ctor, B, Cin, Cout, H, W = case
    torch.manual_seed(hash(name) % 1000 + Cin + Cout)
    m = ctor(Cin, Cout)
    x = torch.randn(B, Cin, H, W)
    with torch.no_grad():
        ref = m(torch.relu(x) if relu_in else x)
        if relu_out:
            ref = torch.relu(ref)
    md = ctor(Cin, Cout).to(dev())
    md.load_state_dict(m.state_dict())
    flags = (1 if relu_in else 0) | (2 if relu_out else 0) | exact
    y = conv_hip.conv(KIND[name], rows(x.to(dev())), md, md.weight, md.bias, Cin, Cout, flags)
    torch.cuda.synchronize()
    close(nchw(y).cpu().numpy(), ref.numpy())


@pytest.mark.parametrize("kind,ctor,Cin,Cout,H,W", [
    ("conv3x3", lambda ci, co: nn.Conv2d(ci, co, 3, 1, 1), 128, 128, 8, 8),
    ("conv4x4s2", lambda ci, co: nn.Conv2d(ci, co, 4, 2, 1), 64, 128, 16, 16),
    ("convT4x4s2", lambda ci, co: nn.ConvTranspose2d(ci, co, 4, 2, 1), 128, 64, 8, 8),
    ("conv1x1", lambda ci, co: nn.Conv2d(ci, co, 1, 1), 128, 64, 8, 8),
])
@pytest.mark.parametrize("wscale", [1.0, 3.0e-5, 7.0e3])
def test_fp16_two_term_scaling_extremes(kind, ctor, Cin, Cout, H, W, wscale):
    """The two-term fp16 path lives on exact power-of-two scales: one per image for the activations, one per layer for
    the weights.  Images of one batch 12 orders of magnitude apart (each far outside fp16's own range), an all-zero
    image, weights scaled far below / above fp16's normal range: every image must still match torch's fp32 conv to the
    usual tolerance RELATIVE TO ITS OWN magnitude."""
    from vqvae_amd import conv_hip
    torch.manual_seed(Cin + Cout + H)
    m = ctor(Cin, Cout)
    with torch.no_grad():
        m.weight.mul_(wscale)
        m.bias.zero_()
    mags = [1.0, 1.0e-6, 3.0e5, 0.0, 2.5e-3, 6.0e4, 1.0e6, 40.0]
    x = torch.randn(len(mags), Cin, H, W) * torch.tensor(mags).view(-1, 1, 1, 1)
    with torch.no_grad():
        ref = m(x)
    md = ctor(Cin, Cout).to(dev())
    md.load_state_dict(m.state_dict())
    y = nchw(conv_hip.conv(KIND[kind], rows(x.to(dev())), md, md.weight, md.bias, Cin, Cout, 0)).cpu().numpy()
    for i, mag in enumerate(mags):
        r = ref[i].numpy()
        np.testing.assert_allclose(y[i], r, atol=1e-5 * max(np.abs(r).max(), 1e-30), rtol=RTOL, err_msg=f"image {i} (x{mag})")
    assert np.all(y[3] == 0.0)


@pytest.mark.parametrize("mag", [1.0, 1.0e-5, 2.0e4])
def test_fp16_two_term_residual_layer_scales(mag):
    """Fused residual layer on the two-term fp16 path: x, the hidden tile and both weight tensors each get their own
    scale; compare with torch per image, images of different magnitude in one batch."""
    from vqvae_amd import conv_hip
    from vqvae_amd.modules import ResidualLayer
    torch.manual_seed(11)
    layer = ResidualLayer(128, 128, 32)
    mags = torch.tensor([mag, 1.0, mag * 300.0, 0.0, mag * 1e-3]).view(-1, 1, 1, 1)
    x = torch.randn(5, 128, 8, 8) * mags
    with torch.no_grad():                                 # models/residual.py:27-29 with its in-place ReLU: relu(x) + block
        xr = torch.relu(x)
        ref = xr + F.conv2d(torch.relu(F.conv2d(xr, layer.res_block[1].weight, padding=1)), layer.res_block[3].weight)
    ld = ResidualLayer(128, 128, 32).to(dev())
    ld.load_state_dict(layer.state_dict())
    for flags in (1, 1 | 8):                              # RELU_IN; the same with the three-term bf16 products
        y = nchw(conv_hip.res_layer(rows(x.to(dev())), ld, flags)).cpu().numpy()
        for i in range(5):
            r = ref[i].numpy()
            np.testing.assert_allclose(y[i], r, atol=1e-5 * max(np.abs(r).max(), 1e-30), rtol=RTOL, err_msg=f"image {i}")


def test_conv_term_products_query():
    from vqvae_amd import _lib
    L = _lib.load()
    assert L.vqvae_conv_term_products(1, 8, 8, 128, 128, 0) == 3         # 8x8 map: two-term fp16
    assert L.vqvae_conv_term_products(0, 16, 16, 64, 128, 0) == 3        # 4x4 s2 on a 16x16 map (tile kernel)
    assert L.vqvae_conv_term_products(1, 56, 56, 128, 128, 0) == 6       # larger maps, per-layer entry: three-term bf16
    assert L.vqvae_conv_term_products(1, 56, 56, 128, 128, 0x100) == 3   # the same layer inside vqvae_forward_f32 (maxima handed over)
    assert L.vqvae_conv_term_products(1, 56, 56, 128, 128, 0x100 | 0x8) == 6
    assert L.vqvae_conv_term_products(1, 8, 8, 128, 128, 8) == 6         # VQVAE_CONV_BF16_SPLIT
    assert L.vqvae_conv_term_products(1, 8, 8, 128, 128, 4) == 1         # VQVAE_CONV_EXACT_FP32


def test_conv_vs_c_oracle():
    from oracle import c_oracle
    from vqvae_amd import conv_hip
    torch.manual_seed(4)
    m = nn.ConvTranspose2d(128, 64, 4, 2, 1).to(dev())
    x = torch.randn(2, 128, 8, 8)
    y = conv_hip.conv(4, rows(x.to(dev())), m, m.weight, m.bias, 128, 64, 2)
    ref = c_oracle.conv_transpose2d(x.numpy(), m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(),
                                    2, 1, relu_out=True)
    close(nchw(y).cpu().numpy(), ref)


@pytest.mark.parametrize("Cin,Cout,B,H,W", [(3, 64, 5, 32, 32), (3, 32, 2, 16, 24), (1, 16, 2, 8, 8), (4, 128, 1, 12, 12),
                                            (3, 64, 2, 64, 64), (3, 32, 1, 256, 256), (1, 96, 3, 32, 32),
                                            # R x TW output tiles that are not whole rows (interior tile borders read real neighbours)
                                            (3, 64, 2, 224, 224), (3, 64, 3, 32, 96), (4, 32, 2, 96, 64)])
def test_conv_in_vs_torch_cpu(Cin, Cout, B, H, W):
    from vqvae_amd.modules import Encoder
    torch.manual_seed(Cin * 10 + Cout)
    enc = Encoder(Cin, 2 * Cout, 0, 8)
    x = torch.randn(B, Cin, H, W)
    with torch.no_grad():
        ref = torch.relu(enc.conv_stack[0](x))
    from vqvae_amd import _lib, conv_hip
    encd = Encoder(Cin, 2 * Cout, 0, 8).to(dev())
    encd.load_state_dict(enc.state_dict())
    c0 = encd.conv_stack[0]
    L = _lib.load()
    xd = x.to(dev())
    p0 = conv_hip._packed(c0, ("conv_in",), c0.weight, lambda: L.vqvae_conv_in_packed_bytes(Cin, Cout),
                          lambda w, buf: L.vqvae_conv_in_pack_f32(w.data_ptr(), Cin, Cout, buf.data_ptr(), None))
    for flags in (2, 2 | 4):                       # RELU_OUT with split-bf16 products (default) / the fp32 MFMA
        y = torch.empty((B, H // 2, W // 2, Cout), device=dev())
        _lib.check(L.vqvae_conv_in_forward_f32(xd.data_ptr(), p0.data_ptr(), c0.bias.data_ptr(), B, H, W, Cin, Cout,
                                               flags, y.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        close(nchw(y).cpu().numpy(), ref.numpy())


@pytest.mark.parametrize("C,Rh,n,B,H,W", [(128, 32, 2, 3, 8, 8), (64, 16, 3, 2, 4, 6), (32, 8, 1, 2, 5, 5)])
def test_residual_stack_quirks_vs_torch_cpu(C, Rh, n, B, H, W):
    """Shared weights + relu(x) on the skip + in-place mutation of the caller's tensor
    (models/residual.py:19,28,44-45)."""
    from oracle import torch_port
    from vqvae_amd.modules import ResidualLayer, ResidualStack
    torch.manual_seed(C + Rh)
    rs = ResidualStack(C, C, Rh, n)
    assert all(l is rs.stack[0] for l in rs.stack)
    assert len(rs.state_dict()) == 2 * n
    x = torch.randn(B, C, H, W)
    w1, w2 = rs.stack[0].res_block[1].weight.detach(), rs.stack[0].res_block[3].weight.detach()
    ref = torch_port.residual_stack(x.clone(), w1, w2, n).numpy()
    rsd = rs.to(dev())
    xd = x.to(dev())
    with torch.no_grad():
        y = rsd(xd)
    close(y.cpu().numpy(), ref)
    # the exact-fp32 MFMA variant of the fused layer kernel
    from vqvae_amd import conv_hip
    t = conv_hip.nchw_to_rows(x.to(dev()))
    for i in range(n):
        t = conv_hip.res_layer(t, rsd.stack[0], (1 if i == 0 else 0) | 2 | 4)
    close(conv_hip.rows_to_nchw(t).cpu().numpy(), ref)
    assert torch.equal(xd.cpu(), torch.relu(x)), "caller's tensor must become relu(x) like upstream"
    # single layer: relu(x) + f(relu(x)), no final relu
    with torch.no_grad():
        xd2 = x.to(dev())
        y1 = rsd.stack[0](xd2)
    t = torch.relu(x)
    ref1 = t + F.conv2d(torch.relu(F.conv2d(t, w1, None, 1, 1)), w2)
    close(y1.cpu().numpy(), ref1.numpy())


@pytest.mark.parametrize("C,B", [(128, 2051), (64, 2049)])
def test_residual_layer_large_batch(C, B):
    """Fused residual layer on a large, odd batch of 8x8 maps (partial last workgroup): a sample of images against the
    CPU restatement, and every image against the same kernel run on 1024-image slices (same bits)."""
    from oracle import torch_port
    from vqvae_amd import conv_hip
    from vqvae_amd.modules import ResidualStack
    torch.manual_seed(C)
    rs = ResidualStack(C, C, 32, 1)
    x = torch.randn(B, C, 8, 8)
    w1, w2 = rs.stack[0].res_block[1].weight.detach(), rs.stack[0].res_block[3].weight.detach()
    pick = [0, 1, 2, 7, 1000, B - 2, B - 1]
    ref = torch_port.residual_stack(x[pick].clone(), w1, w2, 1).numpy()
    rsd = rs.to(dev())
    t = conv_hip.nchw_to_rows(x.to(dev()))
    y = conv_hip.res_layer(t, rsd.stack[0], 1 | 2)                      # relu_in | relu_out == a 1-layer stack
    close(conv_hip.rows_to_nchw(y[pick]).cpu().numpy(), ref)
    halves = [conv_hip.res_layer(t[i:i + 1024].contiguous(), rsd.stack[0], 1 | 2) for i in range(0, B, 1024)]
    assert torch.equal(torch.cat(halves), y), "results must not depend on how the batch is sliced"


def test_convt_out_vs_torch_cpu():
    from vqvae_amd import _lib, conv_hip
    torch.manual_seed(9)
    for Cin, Cout, B, H, W in [(64, 3, 3, 16, 16), (32, 3, 2, 5, 7), (16, 1, 1, 4, 4), (64, 3, 2, 20, 37),
                                  (128, 4, 1, 33, 16), (8, 2, 1, 17, 17)]:
        m = nn.ConvTranspose2d(Cin, Cout, 4, 2, 1)
        x = torch.randn(B, Cin, H, W)
        with torch.no_grad():
            ref = m(x)
        md = nn.ConvTranspose2d(Cin, Cout, 4, 2, 1).to(dev())
        md.load_state_dict(m.state_dict())
        L = _lib.load()
        xr = rows(x.to(dev()))
        p = conv_hip._packed(md, ("convt_out",), md.weight, lambda: L.vqvae_convt_out_packed_bytes(Cin, Cout),
                             lambda w, buf: L.vqvae_convt_out_pack_f32(w.data_ptr(), Cin, Cout, buf.data_ptr(), None))
        for flags in (0, 4):                       # split-bf16 products (default) and VQVAE_CONV_EXACT_FP32
            y = torch.empty((B, Cout, 2 * H, 2 * W), device=dev())
            _lib.check(L.vqvae_convt_out_forward_f32(xr.data_ptr(), p.data_ptr(), md.bias.data_ptr(), B, H, W, Cin, Cout,
                                                     flags, y.data_ptr(), torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            close(y.cpu().numpy(), ref.numpy())


def test_transpose_roundtrip():
    from vqvae_amd import conv_hip
    x = torch.randn(3, 37, 5, 9, device=dev())
    r = conv_hip.nchw_to_rows(x)
    assert torch.equal(r, x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(conv_hip.rows_to_nchw(r), x)


ALIGNED = float(np.float32(1.0) + np.float32(4093.0) * np.float32(2.0 ** -23))      # 1 + 4093 * 2^-23


@pytest.mark.parametrize("kind,ctor,Cin,Cout,H,W", [
    ("conv3x3", lambda ci, co: nn.Conv2d(ci, co, 3, 1, 1, bias=False), 128, 128, 8, 8),       # Kred 1152
    ("conv4x4s2", lambda ci, co: nn.Conv2d(ci, co, 4, 2, 1, bias=False), 64, 128, 16, 16),    # Kred 1024
    ("conv1x1", lambda ci, co: nn.Conv2d(ci, co, 1, 1, bias=False), 128, 64, 8, 8),           # Kred 128
    ("convT3x3", lambda ci, co: nn.ConvTranspose2d(ci, co, 3, 1, 1, bias=False), 64, 128, 8, 8),
    ("convT4x4s2", lambda ci, co: nn.ConvTranspose2d(ci, co, 4, 2, 1, bias=False), 128, 64, 8, 8),
])
def test_fp16_two_term_product_bound_on_aligned_operands(kind, ctor, Cin, Cout, H, W, capsys):
    """The documented error of the two-term fp16 products (include/vqvae_hip.h): x = h1 + h2 + r with
    h1 = fp16(x), h2 = fp16(x - h1), |r| <= 2^-23 |x|; the product keeps h1 g1 + h1 g2 + h2 g1 and drops
    h2 g2 (<= 2^-22 |x w|) + r w + x s (<= 2^-23 |x w| each): at most 2^-21 |x w| per product -- NOT fp32's 2^-24.
    Operands built so that every one of the 64..1152 products of an output errs by that maximum in the SAME direction
    (x = w = 1 + 4093 * 2^-23 up to sign-free powers of two: h2 = 4092 * 2^-23, r = +2^-23), against an fp64 conv:
    the relative error of every output must stay below 2^-21 plus the fp32 accumulation's share, and on these operands it
    must also come out ABOVE 2^-22 (the test would otherwise not be exercising the worst case)."""
    from vqvae_amd import conv_hip
    m = ctor(Cin, Cout)
    B = 3
    with torch.no_grad():
        m.weight.fill_(ALIGNED)
        m.weight.mul_(2.0 ** -7)                                   # exact
    x = torch.full((B, Cin, H, W), ALIGNED) * torch.tensor([1.0, 2.0 ** 9, 2.0 ** -20]).view(-1, 1, 1, 1)
    with torch.no_grad():
        ref = m.double()(x.double())                               # every term positive: ref = sum |x_i w_i|
    md = ctor(Cin, Cout).to(dev())
    md.load_state_dict(m.float().state_dict())
    worst = {}
    for name, flags in (("two-term fp16 (default)", 0), ("three-term bf16", 8), ("fp32 MFMA", 4)):
        y = nchw(conv_hip.conv(KIND[kind], rows(x.to(dev())), md, md.weight, md.bias, Cin, Cout, flags)).cpu().double()
        worst[name] = float(((y - ref).abs() / ref).max())
    with capsys.disabled():
        print(f"\n   {kind}: worst relative output error on aligned operands, in units of 2^-24: " +
              ", ".join(f"{k} {v * 2 ** 24:.2f}" for k, v in worst.items()))
    assert worst["two-term fp16 (default)"] <= 2.0 ** -21 + 2.0 ** -22        # 8 units of 2^-24 + the fp32 accumulation's share
    if kind != "conv1x1":           # (the per-layer 1x1 entry on 8x8 maps comes out at fp32 level on these operands)
        assert worst["two-term fp16 (default)"] >= 2.0 ** -22


@pytest.mark.parametrize("kind,ctor,Cin,Cout,H", [
    (1, lambda ci, co: torch.nn.Conv2d(ci, co, 3, 1, 1), 128, 128, 8),
    (3, lambda ci, co: torch.nn.ConvTranspose2d(ci, co, 3, 1, 1), 64, 128, 8),
    (0, lambda ci, co: torch.nn.Conv2d(ci, co, 4, 2, 1), 64, 128, 16),
    (2, lambda ci, co: torch.nn.Conv2d(ci, co, 1, 1, 0), 128, 128, 8),
])
@pytest.mark.parametrize("flags", [0, 8], ids=["fp16x2", "bf16x3"])
def test_tile_kernel_both_launch_forms(kind, ctor, Cin, Cout, H, flags):
    """conv_tile8_bf3_kernel runs its eight-wave form (one image per wave x all 128 output channels) only when that gives every CU
    a workgroup (B >= 2048 on 256 CUs) and the four-wave form (two channel halves per image) below it (round 4).  The same images
    through both -- one launch of 2304 images against nine launches of 256 -- must agree bit for bit (same operand order per
    accumulator), and a sample of them with torch on the CPU."""
    from vqvae_amd import conv_hip
    dev = torch.device("cuda:0")
    torch.manual_seed(kind * 10 + flags)
    layer = ctor(Cin, Cout)
    B = 2304
    x = torch.randn(B, Cin, H, H)
    xr = x.to(dev).permute(0, 2, 3, 1).contiguous()
    hold = torch.nn.Module()
    big = conv_hip.conv(kind, xr, hold, layer.weight.detach().to(dev), layer.bias.detach().to(dev), Cin, Cout, flags)
    parts = torch.cat([conv_hip.conv(kind, xr[i:i + 256].contiguous(), hold, layer.weight.detach().to(dev), layer.bias.detach().to(dev),
                                     Cin, Cout, flags) for i in range(0, B, 256)])
    assert torch.equal(big, parts)
    with torch.no_grad():
        ref = layer(x[::97]).permute(0, 2, 3, 1)
    got = big[::97].cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-5 * float(ref.abs().max()), rtol=1e-4)
