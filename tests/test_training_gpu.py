"""VQ backward + fused step losses against torch autograd on the CPU oracle.

Oracle: oracle/torch_port.quantize_train -- models/quantizer.py:45-76 with its detach() calls; torch's
autograd over it IS the reference's gradient.  Floating point, so tolerance (stated per check):
  grad_z         rtol 1e-5 (elementwise, two roundings)
  grad_codebook  atol 1e-5 * max|grad| + rtol 1e-4 (segmented sums; the reference sums in MKL's order)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # B, D, H, W, K
    (4, 64, 8, 8, 512),
    (3, 32, 5, 7, 100),
    (2, 128, 4, 4, 1024),
    (16, 64, 8, 8, 37),
    (5, 48, 6, 6, 96),       # round 5: widths outside {32, 64, 128, 256} (the exact-fp32 vector kernel in front of the same backward)
    (3, 7, 4, 5, 50),
]


def _ref_grads(z, cb, beta, g_zq, g_loss):
    from oracle import torch_port
    zc = z.clone().requires_grad_(True)
    cc = cb.clone().requires_grad_(True)
    loss, z_q, _, _, idx = torch_port.quantize_train(zc, cc, beta)
    (g_loss * loss + (z_q * g_zq).sum()).backward()
    return zc.grad, cc.grad, idx


@pytest.mark.parametrize("B,D,H,W,K", CASES)
@pytest.mark.parametrize("rowmajor", [False, True])
def test_vq_backward_vs_autograd(B, D, H, W, K, rowmajor):
    from vqvae_amd import functional as F, training as T
    g = torch.Generator().manual_seed(B * 1000 + K)
    cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
    z = torch.randn(B, D, H, W, generator=g) * 0.05
    g_zq = torch.randn(B, D, H, W, generator=g)
    g_loss = torch.tensor(0.7)
    gz_ref, ge_ref, idx_ref = _ref_grads(z, cb, 0.25, g_zq, g_loss)

    dev = torch.device("cuda:0")
    zd, gd = z.to(dev), g_zq.to(dev)
    if rowmajor:
        zd, gd = zd.permute(0, 2, 3, 1).contiguous(), gd.permute(0, 2, 3, 1).contiguous()
    _, _, _, idx, _ = F.vq_forward(zd, cb.to(dev), 0.25, rowmajor=rowmajor)
    assert torch.equal(idx.cpu(), idx_ref)
    gz, ge = T.vq_backward(zd, cb.to(dev), idx, gd, g_loss.to(dev), 0.25, rowmajor=rowmajor)
    if rowmajor:
        gz = gz.permute(0, 3, 1, 2)
    torch.testing.assert_close(gz.cpu(), gz_ref, rtol=1e-5, atol=1e-9)
    scale = float(ge_ref.abs().max())
    torch.testing.assert_close(ge.cpu(), ge_ref, rtol=1e-4, atol=1e-5 * scale)
    # codes nobody chose get exactly zero gradient
    unused = torch.bincount(idx_ref.view(-1), minlength=K) == 0
    assert (ge.cpu()[unused] == 0).all()


def test_vq_backward_is_bit_reproducible():
    from vqvae_amd import functional as F, training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    cb = ((torch.rand(512, 64, generator=g) * 2 - 1) / 512).to(dev)
    z = (torch.randn(256, 8, 8, 64, generator=g) * 0.05).to(dev)
    _, _, _, idx, _ = F.vq_forward(z, cb, 0.25, rowmajor=True)
    outs = [T.vq_backward(z, cb, idx, None, None, 0.25, rowmajor=True)[1].cpu().numpy().view(np.uint32)
            for _ in range(3)]
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_module_backward_matches_reference_structure():
    """VectorQuantizer.forward under autograd: loss / z_q differentiable, perplexity and the index
    tensors not (SURVEY.md 8b 'Autograd structure'); gradients equal the oracle's."""
    from vqvae_amd.modules import VectorQuantizer
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    vq = VectorQuantizer(512, 64, 0.25).to(dev)
    z_cpu = torch.randn(8, 64, 8, 8) * 0.05
    w_cpu = torch.randn(8, 64, 8, 8)
    z = z_cpu.to(dev).requires_grad_(True)
    loss, z_q, perplexity, min_encodings, idx = vq(z)
    assert loss.requires_grad and z_q.requires_grad
    assert not perplexity.requires_grad and not min_encodings.requires_grad and not idx.requires_grad
    (loss + (z_q * w_cpu.to(dev)).sum()).backward()
    gz_ref, ge_ref, _ = _ref_grads(z_cpu, vq.embedding.weight.detach().cpu(), 0.25, w_cpu, torch.tensor(1.0))
    torch.testing.assert_close(z.grad.cpu(), gz_ref, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(vq.embedding.weight.grad.cpu(), ge_ref, rtol=1e-4,
                               atol=1e-5 * float(ge_ref.abs().max()))
    # straight-through: d(sum z_q)/dz == 1 exactly
    z2 = z_cpu.to(dev).requires_grad_(True)
    vq(z2)[1].sum().backward()
    assert torch.equal(z2.grad, torch.ones_like(z2))


@pytest.mark.parametrize("shape", [(32, 3, 32, 32), (5, 3, 17, 9), (1, 1, 1, 3)])
def test_step_losses_vs_main_py(shape):
    from vqvae_amd import training as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    x, x_hat = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g) * 0.3
    el, pp, var = torch.tensor(0.0123), torch.tensor(37.5), 0.0632
    xh_ref = x_hat.clone().requires_grad_(True)
    el_ref = el.clone().requires_grad_(True)
    recon_ref = torch.mean((xh_ref - x) ** 2) / var          # main.py:75
    loss_ref = recon_ref + el_ref                            # main.py:76
    loss_ref.backward()

    xh = x_hat.to(dev).requires_grad_(True)
    eld = el.to(dev).requires_grad_(True)
    stats = T.step_losses(eld, xh, pp.to(dev), x.to(dev), var)
    assert stats.shape == (3,)
    got = stats.detach().cpu()
    torch.testing.assert_close(got[0], recon_ref.detach(), rtol=2e-6, atol=0)
    torch.testing.assert_close(got[1], loss_ref.detach(), rtol=2e-6, atol=0)
    assert got[2] == pp
    stats[1].backward()
    torch.testing.assert_close(xh.grad.cpu(), xh_ref.grad, rtol=1e-5, atol=1e-12)
    assert float(eld.grad) == 1.0


@pytest.mark.parametrize("backend", ["hip", "torch"])
def test_training_step_runs(backend):
    """main.py:70-79: optimizer steps with the convs on the HIP kernels (forward + backward) or on torch's
    autograd, the quantizer on the HIP forward/backward either way: parameters move, the loss goes down."""
    from vqvae_amd import conv, training as T
    from vqvae_amd.modules import VQVAE
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = VQVAE(128, 32, 2, 512, 64, 0.25).to(dev).train()
    x = torch.randn(16, 3, 32, 32, device=dev)
    conv.set_conv_backend(backend)
    try:
        opt = torch.optim.Adam(model.parameters(), lr=3e-4, amsgrad=True)   # main.py:55
        before = {k: v.detach().clone() for k, v in model.state_dict().items()}
        vals = []
        for _ in range(3):
            opt.zero_grad()
            embedding_loss, x_hat, perplexity = model(x)
            stats = T.step_losses(embedding_loss, x_hat, perplexity, x, 0.06)
            stats[1].backward()
            opt.step()
            vals.append(stats.tolist())
        assert all(np.isfinite(v).all() for v in vals)
        assert vals[-1][0] < vals[0][0]                      # reconstruction error goes down
        moved = [k for k, v in model.state_dict().items() if not torch.equal(v, before[k])]
        assert "vector_quantization.embedding.weight" in moved
        assert "encoder.conv_stack.0.weight" in moved and "decoder.inverse_conv_stack.4.bias" in moved
        assert "encoder.conv_stack.5.stack.0.res_block.1.weight" in moved
    finally:
        conv.set_conv_backend("hip")


# ---------------------------------------------------------------------------------------------------------
# conv backward on the HIP kernels (vqvae_amd/autograd_conv.py) against torch autograd on the CPU.
# Tolerances: data gradients reuse the forward kernels (1e-5 + 1e-4 rel); weight / bias gradients are sums over
# up to B*H*W pixels -> rtol 2e-4 with an absolute floor of 2e-5 * max|grad|.
def _close_grad(got, ref, what):
    scale = float(ref.abs().max()) + 1e-30
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-5 * scale, err_msg=what)


CONV_BWD_CASES = [  # kind, ctor, B, Cin, Cout, H, W
    (0, lambda ci, co: torch.nn.Conv2d(ci, co, 4, 2, 1), 3, 64, 128, 16, 16),
    (0, lambda ci, co: torch.nn.Conv2d(ci, co, 4, 2, 1), 2, 32, 48, 10, 14),
    (1, lambda ci, co: torch.nn.Conv2d(ci, co, 3, 1, 1), 5, 128, 128, 8, 8),
    (1, lambda ci, co: torch.nn.Conv2d(ci, co, 3, 1, 1, bias=False), 2, 32, 16, 7, 9),
    (2, lambda ci, co: torch.nn.Conv2d(ci, co, 1, 1, 0), 3, 128, 64, 8, 8),
    (3, lambda ci, co: torch.nn.ConvTranspose2d(ci, co, 3, 1, 1), 3, 64, 128, 8, 8),
    (3, lambda ci, co: torch.nn.ConvTranspose2d(ci, co, 3, 1, 1), 2, 16, 40, 5, 6),
    (4, lambda ci, co: torch.nn.ConvTranspose2d(ci, co, 4, 2, 1), 3, 128, 64, 8, 8),
    (4, lambda ci, co: torch.nn.ConvTranspose2d(ci, co, 4, 2, 1), 2, 32, 32, 5, 7),
]


@pytest.mark.parametrize("relu_out", [False, True])
@pytest.mark.parametrize("case", CONV_BWD_CASES, ids=lambda c: f"k{c[0]}-{c[3]}to{c[4]}-{c[5]}x{c[6]}")
def test_conv_backward_vs_torch_autograd(case, relu_out):
    from vqvae_amd import autograd_conv as A
    kind, ctor, B, Cin, Cout, H, W = case
    dev = torch.device("cuda:0")
    torch.manual_seed(kind * 100 + Cin + Cout)
    m = ctor(Cin, Cout)
    x = torch.randn(B, Cin, H, W)
    xr = x.clone().requires_grad_(True)
    y = m(xr)
    if relu_out:
        y = torch.relu(y)
    gy = torch.randn_like(y)
    y.backward(gy)
    md = ctor(Cin, Cout).to(dev)
    md.load_state_dict(m.state_dict())
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    yd = A.ConvFn.apply(xd, md.weight, md.bias, md, kind, relu_out)
    np.testing.assert_allclose(yd.permute(0, 3, 1, 2).detach().cpu().numpy(), y.detach().numpy(), atol=1e-5, rtol=1e-4)
    yd.backward(gy.to(dev).permute(0, 2, 3, 1).contiguous())
    _close_grad(xd.grad.permute(0, 3, 1, 2), xr.grad, "grad_x")
    _close_grad(md.weight.grad, m.weight.grad, "grad_w")
    if m.bias is not None:
        _close_grad(md.bias.grad, m.bias.grad, "grad_b")


# the map-resident weight-gradient kernel (8x8 A maps), every wave layout it is built for, more images than image ranges
# (131 images -> 66 ranges of two, the last one of one) against the defining sum in fp64
@pytest.mark.parametrize("k,s,CA,CB", [(3, 1, 128, 128), (3, 1, 64, 128), (3, 1, 32, 128), (3, 1, 128, 32), (4, 2, 128, 64),
                                       (4, 2, 64, 32), (1, 1, 64, 128), (1, 1, 32, 128), (1, 1, 128, 32), (1, 1, 128, 96)])
@pytest.mark.parametrize("exact", [False, True], ids=["fp16x2", "fp32"])
def test_map_resident_weight_gradient_vs_fp64_sum(k, s, CA, CB, exact, monkeypatch):
    from vqvae_amd import autograd_conv as A
    monkeypatch.setattr(A, "WGRAD_EXACT_FP32", exact)
    dev = torch.device("cuda:0")
    torch.manual_seed(k * 1000 + CA + CB)
    B, pad = 131, (0 if k == 1 else 1)
    a = torch.randn(B, 8, 8, CA)
    bt = torch.randn(B, 8 * s, 8 * s, CB)
    ref = torch.nn.grad.conv2d_weight(bt.permute(0, 3, 1, 2).double(), (CA, CB, k, k), a.permute(0, 3, 1, 2).double(),
                                      stride=s, padding=pad)
    got = A.conv_wgrad(a.to(dev), bt.to(dev), k, s, pad)
    _close_grad(got, ref.float(), "grad_w")
    assert torch.equal(got, A.conv_wgrad(a.to(dev), bt.to(dev), k, s, pad))          # fixed-order sums: bit-reproducible


@pytest.mark.parametrize("k,s,CA,CB", [(3, 1, 128, 128), (4, 2, 128, 64), (3, 1, 32, 128), (1, 1, 128, 32)])
def test_two_term_weight_gradient_across_image_scales(k, s, CA, CB):
    """conv_wgrad_map8_h2_kernel carries one power-of-two scale per image and operand tile and rescales its accumulators between
    images: images whose magnitudes differ by up to 10^12 in one range (both directions, both operands), an all-zero image, an
    image below the 2^-60 cut and one channel 10^4 above the rest -- against the defining sum in fp64, relative to each
    (ca, cb) filter's own maximum (2e-5) -- and the same bits from the second call."""
    from vqvae_amd import autograd_conv as A
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(k * 7 + CA)
    B, pad = 37, (0 if k == 1 else 1)
    a = torch.randn(B, 8, 8, CA, generator=g)
    bt = torch.randn(B, 8 * s, 8 * s, CB, generator=g)
    fa = 10.0 ** (torch.rand(B, generator=g) * 12 - 6)
    fb = 10.0 ** (torch.rand(B, generator=g) * 12 - 6)
    fa[3], fb[5] = 0.0, 0.0                          # all-zero operand tiles
    fa[7], fb[7] = 1e-18, 1e-18                      # far below everything summed before it
    fa[0], fb[0] = 1e-6, 1e-6                        # the FIRST image small: the accumulators have to follow upwards
    a *= fa[:, None, None, None]
    bt *= fb[:, None, None, None]
    a[:, :, :, 5] *= 1e4                             # a heterogeneous channel inside every tile
    ref = torch.nn.grad.conv2d_weight(bt.permute(0, 3, 1, 2).double(), (CA, CB, k, k), a.permute(0, 3, 1, 2).double(),
                                      stride=s, padding=pad)
    got = A.conv_wgrad(a.to(dev), bt.to(dev), k, s, pad)
    assert torch.isfinite(got).all()
    err = (got.cpu().double() - ref).abs().amax(dim=(2, 3))
    lim = 2e-5 * ref.abs().amax(dim=(2, 3)) + 1e-30
    # channel 5 of a dominates its tile's scale by 10^4: the other channels keep 22 - 13 bits less ... still inside 2e-5 of the
    # FILTER maximum only for the filters that see channel 5; the others are held to the tile-maximum statement of the header
    tile = ref.abs().amax(dim=(2, 3)).amax(dim=0 if False else 1, keepdim=True)
    assert bool(((err <= lim) | (err <= 2e-7 * tile)).all()), float((err / lim).max())
    assert torch.equal(got, A.conv_wgrad(a.to(dev), bt.to(dev), k, s, pad))
    # a run of EVER SMALLER images (a factor of 10 per image and operand, 10^6 ... 10^-30): the accumulators must not climb after
    # them binade by binade (the first version's rule, relative to the previous image, overflowed here: tests/test_wgrad_scheme_cpu.py)
    a2 = torch.randn(B, 8, 8, CA, generator=g) * (10.0 ** (6 - torch.arange(B, dtype=torch.float32)))[:, None, None, None]
    b2 = torch.randn(B, 8 * s, 8 * s, CB, generator=g) * (10.0 ** (6 - torch.arange(B, dtype=torch.float32)))[:, None, None, None]
    ref2 = torch.nn.grad.conv2d_weight(b2.permute(0, 3, 1, 2).double(), (CA, CB, k, k), a2.permute(0, 3, 1, 2).double(), stride=s, padding=pad)
    got2 = A.conv_wgrad(a2.to(dev), b2.to(dev), k, s, pad)
    assert torch.isfinite(got2).all()
    assert float((got2.cpu().double() - ref2).abs().max()) <= 2e-5 * float(ref2.abs().max())


@pytest.mark.parametrize("C,flags", [(128, 2), (64, 3), (32, 0)])
def test_res_layer_hidden_activation_for_backward(C, flags):
    """vqvae_res_layer_forward_hidden_f32: y bit-identical to the forward-only entry, hidden = relu(W1 * r(x)) of
    models/residual.py:20-23 (fp32 reference on the CPU; two-term fp16 products: 1e-5 + 1e-4 rel)."""
    from vqvae_amd import conv_hip
    from vqvae_amd.modules import ResidualLayer
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    torch.manual_seed(C + flags)
    layer = ResidualLayer(C, C, 32)
    x = torch.randn(7, C, 8, 8)
    ld = ResidualLayer(C, C, 32).to(dev)
    ld.load_state_dict(layer.state_dict())
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous()
    y, hid = conv_hip.res_layer(xd, ld, flags, want_hidden=True)
    assert hid is not None and hid.shape == (7, 8, 8, 32)
    assert torch.equal(y, conv_hip.res_layer(xd, ld, flags))
    r = torch.relu(x) if flags & 1 else x
    ref = torch.relu(F.conv2d(r, layer.res_block[1].weight, None, 1, 1))
    np.testing.assert_allclose(hid.permute(0, 3, 1, 2).cpu().numpy(), ref.detach().numpy(), atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("C,Rh,B,H,W,relu_in,relu_out", [(128, 32, 3, 8, 8, False, True), (128, 32, 2, 8, 8, True, True),
                                                       (64, 16, 2, 5, 7, True, False), (32, 8, 1, 4, 4, False, False)])
def test_res_layer_backward_vs_torch_autograd(C, Rh, B, H, W, relu_in, relu_out):
    from vqvae_amd import autograd_conv as A
    from vqvae_amd.modules import ResidualLayer
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    torch.manual_seed(C + Rh)
    layer = ResidualLayer(C, C, Rh)
    w1, w2 = layer.res_block[1].weight, layer.res_block[3].weight
    x = torch.randn(B, C, H, W)
    xr = x.clone().requires_grad_(True)
    r = torch.relu(xr) if relu_in else xr
    y = r + F.conv2d(torch.relu(F.conv2d(r, w1, None, 1, 1)), w2)
    if relu_out:
        y = torch.relu(y)
    gy = torch.randn_like(y)
    y.backward(gy)
    ld = ResidualLayer(C, C, Rh).to(dev)
    ld.load_state_dict(layer.state_dict())
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    yd = A.ResLayerFn.apply(xd, ld.res_block[1].weight, ld.res_block[3].weight, ld, relu_in, relu_out)
    yd.backward(gy.to(dev).permute(0, 2, 3, 1).contiguous())
    _close_grad(xd.grad.permute(0, 3, 1, 2), xr.grad, "grad_x")
    _close_grad(ld.res_block[1].weight.grad, w1.grad, "grad_w1")
    _close_grad(ld.res_block[3].weight.grad, w2.grad, "grad_w2")


@pytest.mark.parametrize("weights", ["default", "coupled", "independent"])
def test_full_model_backward_on_hip_matches_cpu_reference(weights):
    """loss.backward() of main.py:74-78 entirely on the HIP kernels vs the reference's ops on the CPU.  The decoder
    side sees identical z_q only if no index flips; compare parameter gradients with the flip-free tolerance and
    require identical indices first.  weights: default init, or trained-checkpoint-like per-channel scales (tests/hetero.py:
    the data- and weight-gradient kernels run on two-term fp16 products with per-image scales -- round 4)."""
    from oracle import torch_port
    from tests import hetero
    from vqvae_amd import conv
    from vqvae_amd.modules import VQVAE
    conv.set_conv_backend("hip")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = VQVAE(128, 32, 2, 512, 64, 0.25).train()
    if weights != "default":
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        m.load_state_dict(hetero.rescale_coupled(sd0, 1) if weights == "coupled" else hetero.rescale_independent(sd0, 1))
    x = torch.randn(8, 3, 32, 32)
    # CPU reference with autograd: same parameters as leaves
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    for a, b in (("encoder.conv_stack.5.stack.1.", "encoder.conv_stack.5.stack.0."),
                 ("decoder.inverse_conv_stack.1.stack.1.", "decoder.inverse_conv_stack.1.stack.0.")):
        for k in list(sd):
            if k.startswith(a):
                sd[k] = sd[b + k[len(a):]]                # the aliased pair shares storage upstream
    z_e = torch_port.encode(sd, x.clone(), 2)
    loss_e, z_q, _, _, idx_ref = torch_port.quantize_train(z_e, sd["vector_quantization.embedding.weight"], 0.25)
    x_hat = torch_port.decode(sd, z_q, 2)
    loss_ref = torch.mean((x_hat - x) ** 2) / 0.06 + loss_e
    loss_ref.backward()

    md = m.to(dev)
    embedding_loss, x_hat_d, perplexity = md(x.to(dev))
    loss = torch.mean((x_hat_d - x.to(dev)) ** 2) / 0.06 + embedding_loss
    loss.backward()
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-5)
    worst = {}
    for name, p in md.named_parameters():
        ref = sd[name].grad
        assert ref is not None and p.grad is not None, name
        _close_grad(p.grad, ref, name)
        if ref.dim() == 4:
            # per filter SLICE as well (the per-tensor maximum hides a channel decades below it): every element within
            # 2e-4 relative or 2e-5 of the larger of its dim-0 and dim-1 slice maxima
            err = (p.grad.cpu().double() - ref.double()).abs()
            m0 = ref.abs().amax(dim=(1, 2, 3), keepdim=True).double()
            m1 = ref.abs().amax(dim=(0, 2, 3), keepdim=True).double()
            lim = 2e-4 * ref.abs().double() + 2e-5 * torch.maximum(m0, m1) + 1e-30
            worst[name] = float((err / lim).max())
    print("\n   weight gradients, worst error / per-slice limit:", {k.split(".weight")[0][-28:]: round(v, 3) for k, v in worst.items()})
    assert max(worst.values()) <= 1.0, worst


def test_sub_modules_are_differentiable_on_hip_like_the_reference():
    """Round 5 (VERDICT r4 "missing" 4): models/encoder.py:42-43, models/decoder.py:38-39 and models/residual.py:47-51 are plain
    differentiable modules upstream; here model.encoder(x), model.decoder(z_q) and a ResidualStack on their own record a graph on
    the HIP kernels (autograd_conv's Functions) -- outputs and every parameter / input gradient against the reference's ops under
    torch autograd on the CPU."""
    from oracle import torch_port
    from vqvae_amd import conv
    from vqvae_amd.modules import VQVAE
    conv.set_conv_backend("hip")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = VQVAE(128, 32, 2, 512, 64, 0.25).train()
    x = torch.randn(6, 3, 32, 32)
    zq = 0.1 * torch.randn(6, 64, 8, 8)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    for a, b in (("encoder.conv_stack.5.stack.1.", "encoder.conv_stack.5.stack.0."),
                 ("decoder.inverse_conv_stack.1.stack.1.", "decoder.inverse_conv_stack.1.stack.0.")):
        for k in list(sd):
            if k.startswith(a):
                sd[k] = sd[b + k[len(a):]]
    # CPU reference: the encoder WITHOUT the pre-quantisation conv = the conv stack alone (models/encoder.py:28-43)
    import torch.nn.functional as F
    E = "encoder.conv_stack."
    t = F.relu(F.conv2d(x, sd[E + "0.weight"], sd[E + "0.bias"], stride=2, padding=1))
    t = F.relu(F.conv2d(t, sd[E + "2.weight"], sd[E + "2.bias"], stride=2, padding=1))
    t = F.conv2d(t, sd[E + "4.weight"], sd[E + "4.bias"], padding=1)
    for _ in range(2):                                                   # residual.py:28, :47-51 (out-of-place: same values)
        r = F.relu(t)
        t = r + F.conv2d(F.relu(F.conv2d(r, sd[E + "5.stack.0.res_block.1.weight"], padding=1)), sd[E + "5.stack.0.res_block.3.weight"])
    enc_ref = F.relu(t)
    w_e = torch.randn_like(enc_ref)
    (enc_ref * w_e).sum().backward()
    zq_ref = zq.clone().requires_grad_(True)
    dec_ref = torch_port.decode(sd, zq_ref, 2)
    w_d = torch.randn_like(dec_ref)
    (dec_ref * w_d).sum().backward()

    md = m.to(dev)
    enc = md.encoder(x.to(dev))
    assert enc.requires_grad and enc.shape == enc_ref.shape
    (enc * w_e.to(dev)).sum().backward()
    zq_d = zq.to(dev).requires_grad_(True)
    dec = md.decoder(zq_d)
    (dec * w_d.to(dev)).sum().backward()
    np.testing.assert_allclose(enc.detach().cpu().numpy(), enc_ref.detach().numpy(), atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(dec.detach().cpu().numpy(), dec_ref.detach().numpy(), atol=1e-5, rtol=1e-4)
    _close_grad(zq_d.grad, zq_ref.grad, "grad of the decoder's input")
    for name, p in md.named_parameters():
        if name.startswith(("encoder.", "decoder.")):
            assert p.grad is not None and sd[name].grad is not None, name
            _close_grad(p.grad, sd[name].grad, name)
    # a ResidualStack on its own: relu(x) skip, shared weights, final ReLU -- and the input's gradient
    st = md.encoder.conv_stack[5]
    a = torch.randn(3, 128, 8, 8)
    a_ref = a.clone().requires_grad_(True)
    t = a_ref
    w1, w3 = st.stack[0].res_block[1].weight.detach().cpu(), st.stack[0].res_block[3].weight.detach().cpu()
    for _ in range(2):
        r = F.relu(t)
        t = r + F.conv2d(F.relu(F.conv2d(r, w1, padding=1)), w3)
    out_ref = F.relu(t)
    out_ref.sum().backward()
    a_d = a.to(dev).requires_grad_(True)
    out = st(a_d)
    out.sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), atol=1e-5, rtol=1e-4)
    _close_grad(a_d.grad, a_ref.grad, "grad of the residual stack's input")


@pytest.mark.parametrize("Cin,C0,B,H,W", [(3, 64, 5, 32, 32), (1, 32, 2, 16, 24), (4, 64, 3, 8, 12), (3, 16, 1, 64, 64)])
def test_first_layer_backward_vs_torch_autograd(Cin, C0, B, H, W):
    from vqvae_amd import autograd_conv as A
    dev = torch.device("cuda:0")
    torch.manual_seed(Cin * 7 + C0)
    m = torch.nn.Conv2d(Cin, C0, 4, 2, 1)
    x = torch.randn(B, Cin, H, W)
    y = torch.relu(m(x))
    gy = torch.randn_like(y)
    y.backward(gy)
    md = torch.nn.Conv2d(Cin, C0, 4, 2, 1).to(dev)
    md.load_state_dict(m.state_dict())
    yd = A.ConvInFn.apply(x.to(dev), md.weight, md.bias, md)
    yd.backward(gy.to(dev).permute(0, 2, 3, 1).contiguous())
    _close_grad(md.weight.grad, m.weight.grad, "grad_w")
    _close_grad(md.bias.grad, m.bias.grad, "grad_b")


@pytest.mark.parametrize("C,Cout,B,H,W", [(64, 3, 5, 16, 16), (32, 1, 2, 8, 12), (128, 4, 2, 5, 7), (64, 3, 1, 20, 18)])
def test_last_layer_backward_vs_torch_autograd(C, Cout, B, H, W):
    from vqvae_amd import autograd_conv as A
    dev = torch.device("cuda:0")
    torch.manual_seed(C + Cout)
    m = torch.nn.ConvTranspose2d(C, Cout, 4, 2, 1)
    t = torch.randn(B, C, H, W)
    tr = t.clone().requires_grad_(True)
    y = m(tr)
    gy = torch.randn_like(y)
    y.backward(gy)
    md = torch.nn.ConvTranspose2d(C, Cout, 4, 2, 1).to(dev)
    md.load_state_dict(m.state_dict())
    td = t.to(dev).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    yd = A.ConvTOutFn.apply(td, md.weight, md.bias, md)
    np.testing.assert_allclose(yd.detach().cpu().numpy(), y.detach().numpy(), atol=1e-5, rtol=1e-4)
    yd.backward(gy.to(dev))
    _close_grad(td.grad.permute(0, 3, 1, 2), tr.grad, "grad_t")
    _close_grad(md.weight.grad, m.weight.grad, "grad_w")
    _close_grad(md.bias.grad, m.bias.grad, "grad_b")


# ---- round 4: ReLU masks and skip sums in the data-gradient kernels' epilogues (vqvae_conv_forward_ep_f32) ----------------------
EP_CASES = [  # kind, B, H, W, Cin, Cout
    (1, 5, 8, 8, 128, 128),      # 3x3 on 8x8 maps, four output tiles (conv_tile8_bf3_kernel<4, false, 8>)
    (3, 3, 8, 8, 32, 128),       # conv-transpose 3x3: a residual layer's skip + block gradient
    (4, 3, 8, 8, 128, 64),       # conv-transpose 4x4 s2, four phases (conv_tile8_bf3_kernel<2, false, 4>)
    (0, 3, 16, 16, 64, 128),     # 4x4 s2 over 2x2 input blocks (space-to-depth form)
    (5, 4, 8, 8, 128, 32),       # transposed 1x1, one output tile (conv_igemm_bf3_kernel<1>)
    (2, 2, 5, 7, 64, 64),        # 1x1 on an odd map (generic kernel)
    (1, 2, 12, 12, 32, 64),      # 3x3 on a map that is not 8x8 (generic kernel)
    (1, 3, 8, 8, 32, 60),        # output channels not a multiple of 8: the tile kernel's scalar store path
    (1, 2, 8, 8, 32, 20),        # one ragged channel tile (generic kernel, scalar stores)
]


@pytest.mark.parametrize("case", EP_CASES, ids=lambda c: f"kind{c[0]}-{c[4]}to{c[5]}-{c[2]}x{c[3]}")
@pytest.mark.parametrize("which", ["mask", "addend", "both"])
def test_conv_epilogue_addend_and_mask_bitwise_vs_separate_passes(case, which):
    """y = (mask > 0) ? conv + addend : 0 from the kernel's epilogue is bit for bit conv -> torch add -> vqvae_relu_backward_f32."""
    import torch.nn as nn
    from vqvae_amd import _lib, conv_hip
    kind, B, H, W, Cin, Cout = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(kind * 100 + Cin)
    k = {0: 4, 1: 3, 2: 1, 3: 3, 4: 4, 5: 1}[kind]
    transposed = kind in (3, 4, 5)
    w = (torch.randn((Cin, Cout, k, k) if transposed else (Cout, Cin, k, k), generator=g) * 0.05).to(dev)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    hold = nn.Module()
    y0 = conv_hip.conv(kind, x, hold, w, None, Cin, Cout, 0)
    add = torch.randn(y0.shape, generator=g).to(dev) if which in ("addend", "both") else None
    mask = torch.relu(torch.randn(y0.shape, generator=g)).to(dev) if which in ("mask", "both") else None
    want = y0 if add is None else y0 + add
    if mask is not None:
        want = want * (mask > 0)
    L = _lib.load()
    packed = conv_hip._pack_conv(hold, kind, w, Cin, Cout)
    got = torch.full_like(y0, float("nan"))
    sp = torch.cuda.current_stream(dev).cuda_stream
    rc = L.vqvae_conv_forward_ep_f32(kind, x.data_ptr(), packed.data_ptr(), None, B, H, W, Cin, Cout, 0,
                                     add.data_ptr() if add is not None else None, mask.data_ptr() if mask is not None else None,
                                     got.data_ptr(), sp)
    assert rc == 0, rc                       # every split-product kernel of the per-layer entry has the epilogue
    assert torch.equal(got, want)
    assert torch.equal(conv_hip.conv(kind, x, hold, w, None, Cin, Cout, 0, addend=add, mask=mask), want)
    # the fp32-MFMA form has no such epilogue: the front end runs the separate passes instead (same definition, that form's products)
    y32 = conv_hip.conv(kind, x, hold, w, None, Cin, Cout, 4)
    want32 = y32 if add is None else y32 + add
    if mask is not None:
        want32 = want32 * (mask > 0)
    assert torch.equal(conv_hip.conv(kind, x, hold, w, None, Cin, Cout, 4, addend=add, mask=mask), want32)
    # refusals: in place, misaligned, the fp32-MFMA form
    t = add if add is not None else mask
    assert L.vqvae_conv_forward_ep_f32(kind, x.data_ptr(), packed.data_ptr(), None, B, H, W, Cin, Cout, 0, got.data_ptr(), None,
                                       got.data_ptr(), sp) == -3
    assert L.vqvae_conv_forward_ep_f32(kind, x.data_ptr(), packed.data_ptr(), None, B, H, W, Cin, Cout, 0, t.data_ptr() + 4, None,
                                       got.data_ptr(), sp) == -3
    assert L.vqvae_conv_forward_ep_f32(kind, x.data_ptr(), packed.data_ptr(), None, B, H, W, Cin, Cout, 4, t.data_ptr(), None,
                                       got.data_ptr(), sp) == -3


@pytest.mark.parametrize("C,Cout,B,H,W", [(64, 3, 5, 16, 16), (32, 1, 2, 8, 12), (128, 4, 2, 5, 7)])
def test_last_layer_data_gradient_mask_bitwise(C, Cout, B, H, W):
    """The last layer's data gradient (first-layer kernel on the NCHW gradient) with (t > 0) in its epilogue."""
    import torch.nn as nn
    from vqvae_amd import autograd_conv as A
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + Cout)
    layer = nn.ConvTranspose2d(C, Cout, 4, 2, 1).to(dev)
    t = torch.relu(torch.randn(B, H, W, C, generator=g)).to(dev)
    gout = torch.randn(B, Cout, 2 * H, 2 * W, generator=g).to(dev)
    grads = []
    for fused in (False, True):
        tt = t.clone().requires_grad_(True)
        A.ConvTOutFn.apply(tt, layer.weight, layer.bias, layer, fused).backward(gout)
        grads.append(tt.grad if fused else tt.grad * (t > 0))
    assert torch.equal(grads[0], grads[1])


def test_fused_backward_epilogues_equal_separate_passes_bitwise():
    """The whole training step's parameter gradients with the ReLU masks / skip sums in the data-gradient epilogues are bit for
    bit those of the separate passes (a + b = b + a; a mask is a select), with and without residual layers."""
    from vqvae_amd import autograd_conv as A, conv
    from vqvae_amd.modules import VQVAE
    conv.set_conv_backend("hip")
    dev = torch.device("cuda:0")
    for n_res in (2, 0, 1):
        torch.manual_seed(n_res)
        m = VQVAE(128, 32, n_res, 512, 64, 0.25).train().to(dev)
        x = torch.randn(16, 3, 32, 32, device=dev)
        res = []
        try:
            for fused in (False, True):
                A.FUSE_EPILOGUES = fused
                m.zero_grad(set_to_none=True)
                el, xh, _ = m(x)
                (torch.mean((xh - x) ** 2) / 0.06 + el).backward()
                res.append({k: p.grad.clone() for k, p in m.named_parameters()})
        finally:
            A.FUSE_EPILOGUES = True
        for k in res[0]:
            assert torch.equal(res[0][k], res[1][k]), (n_res, k)


@pytest.mark.parametrize("dims,HW,B", [((64, 16, 1, 64, 32), 16, 6),        # 4x4 latent maps, one residual layer, D = 32
                                       ((32, 32, 3, 100, 64), 24, 3),       # 6x6 maps, three layers, K not a multiple of 32
                                       ((256, 64, 2, 512, 64), 32, 4),      # widths outside the fused residual kernel
                                       ((128, 32, 2, 512, 48), 32, 4)])     # --embedding_dim 48 (round 5)
def test_full_model_backward_other_shapes(dims, HW, B):
    """The autograd path away from main.py's defaults: maps that are not 8x8 (generic conv kernels, per-tap weight-gradient kernel,
    data-gradient epilogues where the kernel has them and separate passes where it has not), other widths and codebooks --
    parameter gradients against the reference's ops on the CPU, flip-free batches only."""
    from oracle import torch_port
    from vqvae_amd import conv
    from vqvae_amd.modules import VQVAE
    conv.set_conv_backend("hip")
    dev = torch.device("cuda:0")
    h, rh, nres, K, D = dims
    torch.manual_seed(11)
    m = VQVAE(h, rh, nres, K, D, 0.25).train()
    x = torch.randn(B, 3, HW, HW)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    for i in range(1, nres):                                   # the stack's layers share storage upstream
        for side in ("encoder.conv_stack.5.stack.", "decoder.inverse_conv_stack.1.stack."):
            for k in list(sd):
                if k.startswith(f"{side}{i}."):
                    sd[k] = sd[f"{side}0." + k[len(f"{side}{i}."):]]
    z_e = torch_port.encode(sd, x.clone(), nres)
    loss_e, z_q, _, _, idx_ref = torch_port.quantize_train(z_e, sd["vector_quantization.embedding.weight"], 0.25)
    x_hat = torch_port.decode(sd, z_q, nres)
    loss_ref = torch.mean((x_hat - x) ** 2) / 0.06 + loss_e
    loss_ref.backward()
    md = m.to(dev)
    embedding_loss, x_hat_d, _ = md(x.to(dev))
    loss = torch.mean((x_hat_d - x.to(dev)) ** 2) / 0.06 + embedding_loss
    loss.backward()
    idx_d = md.encode(x.to(dev)).cpu().view(-1)
    if not torch.equal(idx_d, idx_ref.view(-1)):
        pytest.skip("an index flip on this batch: the decoder sides differ by construction")
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-5)
    for name, p in md.named_parameters():
        ref = sd[name].grad
        assert ref is not None and p.grad is not None, name
        _close_grad(p.grad, ref, name)
