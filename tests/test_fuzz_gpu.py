"""Seeded random-shape sweep through the C ABI (-m gpu): the quantizer against the C oracle (bit-exact
indices / z_q), every conv kind and the fused residual layer against torch's CPU fp32 ops
(|y - y_ref| <= 1e-5 + 1e-4 |y_ref|).  Shapes are drawn inside the documented support of each entry point,
deliberately hitting ragged tails (pixel counts not a multiple of 32/64/128/256, odd maps, K not a multiple
of 32, partial last workgroups) and both the tile-resident (8x8 / 16x16) and the generic kernels."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _rng(seed):
    return np.random.default_rng(seed)


VQ_SEEDS = list(range(48))


@pytest.mark.parametrize("seed", VQ_SEEDS)
def test_vq_random_shapes_bit_exact(seed):
    from oracle import c_oracle
    from vqvae_amd import functional as Fh
    r = _rng(1000 + seed)
    D = int(r.choice([32, 64, 64, 64, 128, 256]))
    K = int(r.integers(1, 700)) if seed % 3 else int(r.choice([1, 31, 32, 33, 512, 513, 1024]))
    B, H, W = int(r.integers(1, 6)), int(r.integers(1, 12)), int(r.integers(1, 12))
    kind = seed % 4
    g = torch.Generator().manual_seed(seed)
    if kind == 0:                                   # reference init scale: tiny margins
        cb = (torch.rand(K, D, generator=g) * 2 - 1) / max(K, 2)
        z = torch.randn(B, D, H, W, generator=g) * 0.07
    elif kind == 1:                                 # unit normal
        cb, z = torch.randn(K, D, generator=g), torch.randn(B, D, H, W, generator=g)
    elif kind == 2:                                 # rows near codes (trained-like) with duplicated codes
        cb = torch.randn(K, D, generator=g)
        if K > 3:
            cb[K // 2] = cb[0]
        sel = torch.randint(0, K, (B * H * W,), generator=g)
        z = (cb[sel] + 0.05 * torch.randn(B * H * W, D, generator=g)).view(B, H, W, D).permute(0, 3, 1, 2).contiguous()
    else:                                           # wide dynamic range
        cb = torch.randn(K, D, generator=g) * torch.logspace(-3, 2, K).unsqueeze(1)
        z = torch.randn(B, D, H, W, generator=g) * 10
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    for rowmajor in (False, True):
        zd = z.to(dev())
        if rowmajor:
            zd = zd.permute(0, 2, 3, 1).contiguous()
        loss, z_q, ppl, idx, hist = Fh.vq_forward(zd, cb.to(dev()), 0.25, rowmajor=rowmajor)
        if rowmajor:
            z_q = z_q.permute(0, 3, 1, 2)
        assert np.array_equal(idx.cpu().numpy(), ref["idx"]), f"K={K} D={D} N={B*H*W} kind={kind}"
        assert np.array_equal(z_q.contiguous().cpu().numpy().view(np.uint32), ref["z_q"].view(np.uint32))
        assert np.array_equal(hist.cpu().numpy(), ref["hist"])
        np.testing.assert_allclose(loss.item(), ref["loss"], rtol=2e-6)
        np.testing.assert_allclose(ppl.item(), ref["perplexity"], rtol=1e-5)

@pytest.mark.parametrize("seed", list(range(40)))
def test_vq_random_widths_and_dead_code_clusters_bit_exact(seed):
    """Round 6's two new quantizer paths under random shapes, against the C oracle bit for bit:
      * ANY embedding width 1 <= D <= 256 on the fp32 matrix cores (vq_anyd_kernel; even seeds) and on round 5's vector kernel
        (VQVAE_VQ_BF16_FILTER), ragged K / row counts, NCHW and row-major, a NaN / Inf row now and then;
      * D = 64 codebooks with a CLUSTER of near-identical codes of random size (a trained checkpoint's dead codes; odd seeds): rows at the
        cluster overflow the exact part's task table and are taken by the whole wave -- every launch form."""
    from oracle import c_oracle
    from vqvae_amd import functional as Fh
    r = _rng(7000 + seed)
    g = torch.Generator().manual_seed(7000 + seed)
    if seed % 2 == 0:
        D = int(r.choice([1, 2, 3, 5, 7, 8, 9, 12, 17, 24, 31, 33, 40, 48, 63, 65, 72, 96, 100, 127, 129, 160, 200, 255]))
        K = int(r.choice([1, 2, 31, 33, 64, 100, 257, 512, 700, 1025]))
        B, H, W = int(r.integers(1, 5)), int(r.integers(1, 11)), int(r.integers(1, 11))
        scale = float(r.choice([0.07, 1.0, 30.0]))
        cb = torch.randn(K, D, generator=g) * (1.0 / K if scale < 1 else 1.0)
        z = torch.randn(B, D, H, W, generator=g) * scale
        if seed % 6 == 0 and B * H * W > 3:
            zr = z.permute(0, 2, 3, 1).reshape(-1, D)
            zr[1, D // 2] = float("nan")
            zr[2, 0] = float("inf")
            z = zr.view(B, H, W, D).permute(0, 3, 1, 2).contiguous()
        forms = [dict(), dict(bf16_filter=True)]
    else:
        D, K = 64, int(r.choice([96, 256, 512, 512, 1024]))
        m = int(r.choice([3, 20, 63, 64, 65, 130, K // 2, K - 8]))                 # cluster size
        m = max(2, min(m, K - 1))
        cb = torch.randn(K, D, generator=g) * float(r.choice([1.0, 8.0]))
        centre = torch.randn(D, generator=g) * float(r.choice([0.0, 1.0]))           # at the origin (dead codes) or anywhere
        where = torch.randperm(K, generator=g)[:m]
        cb[where] = centre + torch.randn(m, D, generator=g) * float(r.choice([1e-7, 1e-4, 2e-3]))
        B, H, W = int(r.integers(1, 40)), 8, 8
        n = B * 64
        zr = cb[torch.randint(0, K, (n,), generator=g)] + 0.05 * torch.randn(n, D, generator=g)
        hit = torch.rand(n, generator=g) < float(r.choice([0.01, 0.2, 0.9]))
        zr[hit] = centre + torch.randn(n, D, generator=g)[hit] * float(r.choice([1e-5, 1e-2, 0.3]))
        z = zr.view(B, H, W, D).permute(0, 3, 1, 2).contiguous()
        forms = [dict(), dict(form=8), dict(form=16), dict(bf16_filter=True)]
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    for kw in forms:
        for rowmajor in (False, True):
            zd = z.to(dev())
            if rowmajor:
                zd = zd.permute(0, 2, 3, 1).contiguous()
            loss, z_q, ppl, idx, hist = Fh.vq_forward(zd, cb.to(dev()), 0.25, rowmajor=rowmajor, **kw)
            if rowmajor:
                z_q = z_q.permute(0, 3, 1, 2)
            what = f"seed={seed} K={K} D={D} N={B * H * W} {kw} rowmajor={rowmajor}"
            assert np.array_equal(idx.cpu().numpy(), ref["idx"]), what
            got, want = z_q.contiguous().cpu().numpy(), ref["z_q"]
            assert np.array_equal(np.isnan(got), np.isnan(want)), what
            ok = ~np.isnan(want)
            assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32)), what
            assert np.array_equal(hist.cpu().numpy(), ref["hist"]), what



CONV_SEEDS = list(range(90))


@pytest.mark.parametrize("seed", CONV_SEEDS)
def test_conv_random_shapes_vs_torch_cpu(seed):
    from vqvae_amd import conv_hip
    r = _rng(2000 + seed)
    kind = int(r.integers(0, 5))
    Cin = int(r.choice([4, 8, 16, 32, 64, 96, 128]))
    Cout = int(r.choice([3, 8, 32, 48, 64, 128]))
    B = int(r.integers(1, 7))
    if seed % 3 == 0:                               # the benchmark's map sizes: tile-resident kernels
        H = W = 16 if kind == 0 else 8
    else:
        H, W = int(r.integers(1, 8)) * 2, int(r.integers(1, 11)) * 2 if kind == 0 else int(r.integers(1, 19))
        if kind != 0:
            H = int(r.integers(1, 15))
    bias = bool(r.integers(0, 2))
    relu_in, relu_out = bool(r.integers(0, 2)), bool(r.integers(0, 2))
    torch.manual_seed(seed)
    m = [lambda: nn.Conv2d(Cin, Cout, 4, 2, 1, bias=bias), lambda: nn.Conv2d(Cin, Cout, 3, 1, 1, bias=bias),
         lambda: nn.Conv2d(Cin, Cout, 1, 1, 0, bias=bias), lambda: nn.ConvTranspose2d(Cin, Cout, 3, 1, 1, bias=bias),
         lambda: nn.ConvTranspose2d(Cin, Cout, 4, 2, 1, bias=bias)][kind]()
    x = torch.randn(B, Cin, H, W)
    with torch.no_grad():
        ref = m(torch.relu(x) if relu_in else x)
        if relu_out:
            ref = torch.relu(ref)
    md = m.to(dev())
    for exact in (False, True):
        flags = (1 if relu_in else 0) | (2 if relu_out else 0) | (4 if exact else 0)
        y = conv_hip.conv(kind, x.to(dev()).permute(0, 2, 3, 1).contiguous(), md, md.weight, md.bias, Cin, Cout, flags)
        np.testing.assert_allclose(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), atol=1e-5, rtol=1e-4,
                                   err_msg=f"kind={kind} B={B} {Cin}->{Cout} {H}x{W} exact={exact}")


@pytest.mark.parametrize("seed", list(range(30)))
def test_res_layer_random_shapes_vs_torch_cpu(seed):
    from vqvae_amd import conv_hip
    from vqvae_amd.modules import ResidualLayer
    r = _rng(3000 + seed)
    C = int(r.choice([32, 64, 128]))
    Rh = int(r.choice([4, 8, 16, 32]))
    B = int(r.integers(1, 10))
    H, W = (8, 8) if seed % 2 == 0 else (int(r.integers(1, 13)), int(r.integers(1, 13)))
    relu_in, relu_out = bool(r.integers(0, 2)), bool(r.integers(0, 2))
    torch.manual_seed(seed)
    layer = ResidualLayer(C, C, Rh)
    x = torch.randn(B, C, H, W)
    w1, w2 = layer.res_block[1].weight.detach(), layer.res_block[3].weight.detach()
    t = torch.relu(x) if relu_in else x
    ref = t + F.conv2d(torch.relu(F.conv2d(t, w1, None, 1, 1)), w2)
    if relu_out:
        ref = torch.relu(ref)
    ld = layer.to(dev())
    for exact in (False, True):
        flags = (1 if relu_in else 0) | (2 if relu_out else 0) | (4 if exact else 0)
        y = conv_hip.res_layer(x.to(dev()).permute(0, 2, 3, 1).contiguous(), ld, flags)
        np.testing.assert_allclose(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), atol=1e-5, rtol=1e-4,
                                   err_msg=f"C={C} Rh={Rh} B={B} {H}x{W} exact={exact}")


@pytest.mark.parametrize("seed", list(range(16)))
def test_first_and_last_layer_random_shapes_vs_torch_cpu(seed):
    """conv_in (NCHW image -> row-major) and convT_out (row-major -> NCHW image), incl. the LDS-staged row-band
    path (32x32, 64x64, 256-wide) and the halo-tiled last layer (maps larger than 16x16)."""
    from vqvae_amd import _lib, conv_hip
    r = _rng(4000 + seed)
    L = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    Cin = int(r.choice([1, 3, 3, 4]))
    C0 = int(r.choice([8, 16, 32, 64, 96, 128]))
    B = int(r.integers(1, 5))
    H, W = [(32, 32), (64, 64), (16, 256), (int(r.integers(1, 20)) * 2, int(r.integers(1, 20)) * 2)][seed % 4]
    torch.manual_seed(seed)
    c0 = nn.Conv2d(Cin, C0, 4, 2, 1)
    x = torch.randn(B, Cin, H, W)
    with torch.no_grad():
        ref = torch.relu(c0(x))
    cd = c0.to(dev())
    p0 = conv_hip._packed(cd, ("conv_in",), cd.weight, lambda: L.vqvae_conv_in_packed_bytes(Cin, C0),
                          lambda w, buf: L.vqvae_conv_in_pack_f32(w.data_ptr(), Cin, C0, buf.data_ptr(), None))
    y = torch.empty((B, H // 2, W // 2, C0), device=dev())
    _lib.check(L.vqvae_conv_in_forward_f32(x.to(dev()).data_ptr(), p0.data_ptr(), cd.bias.data_ptr(), B, H, W, Cin, C0, 2,
                                           y.data_ptr(), st))
    np.testing.assert_allclose(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), atol=1e-5, rtol=1e-4,
                               err_msg=f"conv_in {Cin}->{C0} B={B} {H}x{W}")
    # last layer on a (B, h, w, Ci) map
    Ci = int(r.choice([8, 32, 64, 128]))
    Co = int(r.choice([1, 3, 3, 4]))
    h, w = [(16, 16), (8, 8), (int(r.integers(1, 40)), int(r.integers(1, 40))), (17, 33)][seed % 4]
    d4 = nn.ConvTranspose2d(Ci, Co, 4, 2, 1)
    t = torch.randn(B, Ci, h, w)
    with torch.no_grad():
        ref2 = d4(t)
    dd = d4.to(dev())
    p4 = conv_hip._packed(dd, ("convt_out",), dd.weight, lambda: L.vqvae_convt_out_packed_bytes(Ci, Co),
                          lambda wt, buf: L.vqvae_convt_out_pack_f32(wt.data_ptr(), Ci, Co, buf.data_ptr(), None))
    xh = torch.empty((B, Co, 2 * h, 2 * w), device=dev())
    td = t.to(dev()).permute(0, 2, 3, 1).contiguous()
    _lib.check(L.vqvae_convt_out_forward_f32(td.data_ptr(), p4.data_ptr(), dd.bias.data_ptr(), B, h, w, Ci, Co, 0,
                                             xh.data_ptr(), st))
    np.testing.assert_allclose(xh.cpu().numpy(), ref2.numpy(), atol=1e-5, rtol=1e-4,
                               err_msg=f"convT_out {Ci}->{Co} B={B} {h}x{w}")


@pytest.mark.parametrize("seed", list(range(12)))
def test_model_random_configs_stagewise_vs_oracle(seed):
    """Random VQVAE configurations (widths, residual depth incl. 0, K, D, image size), each stage fed the
    live oracle's input bits (oracle/torch_port.py = the reference's ATen ops) as in test_model_gpu.py."""
    from oracle import torch_port
    from vqvae_amd import conv, conv_hip
    from vqvae_amd.modules import VQVAE
    conv.set_conv_backend("hip")
    r = _rng(5000 + seed)
    h = int(r.choice([64, 128]))
    rh = int(r.choice([8, 16, 32]))
    nl = int(r.integers(0, 4))
    K = int(r.integers(2, 600))
    D = int(r.choice([32, 64, 128]))
    B = int(r.integers(1, 6))
    H, W = (32, 32) if seed % 3 == 0 else (int(r.integers(2, 13)) * 4, int(r.integers(2, 13)) * 4)
    torch.manual_seed(seed)
    m = VQVAE(h, rh, nl, K, D, 0.25).eval()
    x = torch.randn(B, 3, H, W)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    if nl == 0:     # torch_port indexes stack.0 weights unconditionally; an empty stack has none
        sd["encoder.conv_stack.5.stack.0.res_block.1.weight"] = sd["encoder.conv_stack.5.stack.0.res_block.3.weight"] = None
        sd["decoder.inverse_conv_stack.1.stack.0.res_block.1.weight"] = sd["decoder.inverse_conv_stack.1.stack.0.res_block.3.weight"] = None
    with torch.no_grad():
        z_e_ref = torch_port.encode(sd, x.clone(), nl)
        loss_ref, z_q_ref, ppl_ref, _, idx_ref = torch_port.quantize(z_e_ref, sd["vector_quantization.embedding.weight"], 0.25)
        x_hat_ref = torch_port.decode(sd, z_q_ref.clone(), nl)
    md = m.to(dev())
    tag = f"h={h} rh={rh} nl={nl} K={K} D={D} B={B} {H}x{W}"
    with torch.no_grad():
        z_e = conv_hip.encoder_forward(md.encoder, x.to(dev()), md.pre_quantization_conv)
        np.testing.assert_allclose(z_e.permute(0, 3, 1, 2).cpu().numpy(), z_e_ref.numpy(), atol=2e-6, rtol=1e-5, err_msg=tag)
        loss, z_q, ppl, _, idx = md.vector_quantization(z_e_ref.to(dev()))
        assert torch.equal(idx.cpu(), idx_ref), tag
        assert torch.equal(z_q.cpu(), z_q_ref), tag
        np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=2e-6, err_msg=tag)
        x_hat = md.decoder(z_q_ref.to(dev()))
        np.testing.assert_allclose(x_hat.cpu().numpy(), x_hat_ref.numpy(), atol=1e-5, rtol=1e-4, err_msg=tag)
        out = md(x.to(dev()))
        assert out[1].shape == x.shape and torch.isfinite(out[1]).all()


@pytest.mark.parametrize("seed", list(range(24)))
def test_conv_backward_random_shapes_vs_torch_autograd(seed):
    """Data, weight and bias gradients of every conv kind on random shapes (tile-resident and generic kernels,
    ragged channel counts, pixel counts that are not multiples of the 32-pixel wgrad block)."""
    from vqvae_amd import autograd_conv as A
    r = _rng(6000 + seed)
    kind = int(r.integers(0, 5))
    Cin = int(r.choice([4, 8, 16, 32, 64, 128]))
    Cout = int(r.choice([4, 8, 24, 32, 64, 128]))          # Cout % 4: the data gradient reads it as its input
    B = int(r.integers(1, 6))
    if seed % 3 == 0:
        H = W = 16 if kind == 0 else 8
    else:
        H = int(r.integers(1, 8)) * 2 if kind == 0 else int(r.integers(1, 13))
        W = int(r.integers(1, 8)) * 2 if kind == 0 else int(r.integers(1, 13))
    relu_out = bool(r.integers(0, 2))
    torch.manual_seed(seed)
    ctor = [lambda: nn.Conv2d(Cin, Cout, 4, 2, 1), lambda: nn.Conv2d(Cin, Cout, 3, 1, 1), lambda: nn.Conv2d(Cin, Cout, 1),
            lambda: nn.ConvTranspose2d(Cin, Cout, 3, 1, 1), lambda: nn.ConvTranspose2d(Cin, Cout, 4, 2, 1)][kind]
    m = ctor()
    x = torch.randn(B, Cin, H, W)
    xr = x.clone().requires_grad_(True)
    y = m(xr)
    if relu_out:
        y = torch.relu(y)
    gy = torch.randn_like(y)
    y.backward(gy)
    md = ctor().to(dev())
    md.load_state_dict(m.state_dict())
    xd = x.to(dev()).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    yd = A.ConvFn.apply(xd, md.weight, md.bias, md, kind, relu_out)
    yd.backward(gy.to(dev()).permute(0, 2, 3, 1).contiguous())
    tag = f"kind={kind} B={B} {Cin}->{Cout} {H}x{W} relu={relu_out}"
    for got, ref, what in ((xd.grad.permute(0, 3, 1, 2), xr.grad, "grad_x"), (md.weight.grad, m.weight.grad, "grad_w"),
                           (md.bias.grad, m.bias.grad, "grad_b")):
        scale = float(ref.abs().max()) + 1e-30
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-5 * scale, err_msg=f"{what} {tag}")
