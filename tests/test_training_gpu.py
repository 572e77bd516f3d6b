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


def test_training_step_runs_on_hip_quantizer():
    """main.py:70-79 with the HIP quantizer (forward + backward) and torch convs: parameters move,
    the loss is finite, and the HIP conv backend refuses to record a graph."""
    from vqvae_amd import conv, training as T
    from vqvae_amd._lib import VqvaeHipError
    from vqvae_amd.modules import VQVAE
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = VQVAE(128, 32, 2, 512, 64, 0.25).to(dev).train()
    x = torch.randn(16, 3, 32, 32, device=dev)
    with pytest.raises(VqvaeHipError):
        model(x)                                             # "hip" convs: forward-only
    conv.set_conv_backend("torch")
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
    finally:
        conv.set_conv_backend("hip")
