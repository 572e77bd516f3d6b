"""GatedPixelCNN prior: oracle pinned to the real reference's logits (CPU), the HIP mirror
pinned to the same goldens (GPU).  Goldens: oracle/gen_golden_pixelcnn.py ran pixelcnn/models.py::GatedPixelCNN
(imported from the reference tree in the build container) on seeded inputs -> tests/golden/pixelcnn_cases.npz.

Floating point: the HIP path forms conv products from exact three-term bf16 splits (fp32-grade) and evaluates
tanh / sigmoid with the device math library, the reference with the CPU's: logits agree to atol 2e-4 + rtol 1e-4
over 15 gated layers (observed ~1e-5); the argmax of every position must agree wherever the top-2 margin exceeds that.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"k512_dim64_l15": (512, 64, 15, 10, 4, 8, 8), "k64_dim32_l3": (64, 32, 3, 5, 3, 6, 6)}


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "pixelcnn_cases.npz"))


def _inputs(name):
    K, dim, nl, ncls, B, H, W = CASES[name]
    g = torch.Generator().manual_seed(77 + len(name))
    x = torch.randint(0, K, (B, H, W), generator=g)
    label = torch.randint(0, ncls, (B,), generator=g)
    return x, label


def _build(name):
    from vqvae_amd.pixelcnn import GatedPixelCNN
    K, dim, nl, ncls, B, H, W = CASES[name]
    torch.manual_seed(0)
    m = GatedPixelCNN(K, dim, nl, ncls).eval()
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n_))) * 0.05)
    return m


def _sd_sha(sd):
    return np.frombuffer(hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k in sorted(sd))).digest()[:8], dtype=np.uint8)


@pytest.mark.parametrize("name", list(CASES))
def test_mirror_state_dict_is_the_reference_layout(name, golden):
    m = _build(name)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert list(sd.keys()) == list(golden[f"{name}/sd_keys"])
    assert np.array_equal(_sd_sha(sd), golden[f"{name}/sd_sha"]), "default init differs from the reference's"


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_port_matches_reference_logits_bitwise(name, golden):
    from oracle import pixelcnn_port
    m = _build(name)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x, label = _inputs(name)
    with torch.no_grad():
        logits = pixelcnn_port.forward(sd, x, label, CASES[name][2])
    assert np.array_equal(logits.numpy().view(np.uint32), golden[f"{name}/logits"].view(np.uint32))
    # make_causal mutated the mask-A weights in place, like the reference
    assert float(sd["layers.0.vert_stack.weight"][:, :, -1].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_forward_matches_reference_logits(name, golden):
    dev = torch.device("cuda:0")
    m = _build(name).to(dev)
    x, label = _inputs(name)
    logits = m(x.to(dev), label.to(dev))
    ref = golden[f"{name}/logits"]
    assert logits.shape == ref.shape and logits.is_contiguous()
    got = logits.cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=2e-4, rtol=1e-4)
    # argmax per position agrees wherever the reference's top-2 margin is above the tolerance
    srt = np.sort(ref, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-3
    assert np.array_equal(got.argmax(1)[clear], ref.argmax(1)[clear])
    # mask-A weights were zeroed in place, as upstream
    assert float(m.layers[0].vert_stack.weight.detach()[:, :, -1].abs().max()) == 0.0
    with pytest.raises(Exception):
        m(x, label)                                       # CPU tensors: no fallback


@pytest.mark.gpu
def test_hip_layer_boundary_and_generate():
    """GatedMaskedConv2d at its NCHW boundary vs the oracle's ops; generate() returns valid, causal samples:
    position (i, j) depends only on earlier positions, so re-running the forward on the finished sample
    reproduces the distribution each position was drawn from (checked through determinism of the logits)."""
    from oracle import pixelcnn_port
    dev = torch.device("cuda:0")
    m = _build("k64_dim32_l3").to(dev)
    K, dim, nl, ncls, B, H, W = CASES["k64_dim32_l3"]
    torch.manual_seed(5)
    label = torch.randint(0, ncls, (6,), device=dev)
    s = m.generate(label, shape=(6, 6), batch_size=6)
    torch.manual_seed(5)
    torch.randint(0, ncls, (6,), device=dev)               # same generator position as before the first call
    sg = m.generate(label, shape=(6, 6), batch_size=6, use_graph=True)
    assert torch.equal(s, sg), "hipGraph replay must sample exactly what eager launches sample"
    assert s.shape == (6, 6, 6) and s.dtype == torch.int64 and int(s.min()) >= 0 and int(s.max()) < K
    # causality: changing a LATER pixel must not change the logits of an earlier position
    s2 = s.clone()
    s2[:, 4, 3] = (s2[:, 4, 3] + 1) % K
    l1, l2 = m(s, label), m(s2, label)
    assert torch.equal(l1[:, :, :4, :], l2[:, :, :4, :]) and torch.equal(l1[:, :, 4, :4], l2[:, :, 4, :4])
    assert not torch.equal(l1[:, :, 4, 4:], l2[:, :, 4, 4:]) or not torch.equal(l1[:, :, 5, :], l2[:, :, 5, :])
    # and the whole thing agrees with the oracle on the generated sample
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = pixelcnn_port.forward(sd, s.cpu(), label.cpu(), nl)
    np.testing.assert_allclose(l1.cpu().numpy(), ref.numpy(), atol=2e-4, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,Cout,taps", [
    (5, 8, 8, 64, 128, [(ky - 1, kx - 1) for ky in range(2) for kx in range(3)]),      # vertical stack, k = 3: tile-resident kernel
    (5, 8, 8, 64, 128, [(0, kx - 1) for kx in range(2)]),                               # horizontal stack, k = 3
    (3, 8, 8, 32, 64, [(0, kx - 3) for kx in range(4)]),                                # horizontal stack, k = 7
    (2, 6, 6, 32, 64, [(ky - 1, kx - 1) for ky in range(2) for kx in range(3)]),       # a map the generic kernel takes
    (2, 5, 7, 16, 24, [(-2, 3), (0, 0), (1, -1)]),                                      # an arbitrary list, odd shapes
    (3, 8, 8, 64, 128, [(ky - 3, kx - 3) for ky in range(4) for kx in range(7)]),      # the first layer's 4 x 7 stack: two slices of 14
    (2, 6, 6, 32, 64, [(ky - 3, kx - 3) for ky in range(4) for kx in range(7)]),
])
def test_tap_list_conv_vs_shifted_sum(B, H, W, Cin, Cout, taps):
    """vqvae_conv_taps_forward_f32 against the defining sum y[b,y,x,:] = b + sum_t W[:, :, t] x[b, y + dy_t, x + dx_t, :] (zero outside
    the map), fp64 on the CPU; fp32-grade products: atol 1e-5 * max|y| + rtol 1e-4."""
    import torch.nn as nn
    from vqvae_amd import conv_hip
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(len(taps) * 100 + Cin)
    n = len(taps)
    w = torch.randn(Cout, Cin, 1, n, generator=g) * 0.1           # ((1, n): any (kh, kw) with kh * kw = n is the same memory)
    bias = torch.randn(Cout, generator=g)
    x = torch.randn(B, H, W, Cin, generator=g)
    ref = bias.double().expand(B, H, W, Cout).clone()
    xd = x.double()
    for t, (dy, dx) in enumerate(taps):
        sh = torch.zeros_like(xd)
        ys, ye = max(0, -dy), min(H, H - dy)
        xs, xe = max(0, -dx), min(W, W - dx)
        if ys < ye and xs < xe:
            sh[:, ys:ye, xs:xe] = xd[:, ys + dy:ye + dy, xs + dx:xe + dx]
        ref += sh @ w[:, :, 0, t].double().T
    hold = nn.Module()
    got = conv_hip.conv_taps(x.to(dev), hold, w.to(dev), bias.to(dev), taps).cpu().double()
    lim = 1e-5 * float(ref.abs().max()) + 1e-4 * ref.abs()
    assert bool(((got - ref).abs() <= lim).all()), float(((got - ref).abs() / lim).max())
    L = __import__("vqvae_amd._lib", fromlist=["load"]).load()
    assert L.vqvae_conv_taps_packed_bytes(17, Cin, Cout) == 0 and L.vqvae_conv_taps_packed_bytes(0, Cin, Cout) == 0


@pytest.mark.gpu
def test_im2col_entry_agrees_with_the_tap_list_conv():
    """vqvae_im2col_rows_f32 (round 2's path for the masked convs, still part of the C ABI) + the 1x1 GEMM against the tap-list
    convolution that replaced it: the same sums in a different order."""
    import torch.nn as nn
    from vqvae_amd import conv_hip, pixelcnn
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    taps = [(ky - 1, kx - 1) for ky in range(2) for kx in range(3)]
    B, H, W, Cin, Cout = 4, 8, 8, 64, 128
    w = (torch.randn(Cout, Cin, 2, 3, generator=g) * 0.1).to(dev)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    cols = pixelcnn._im2col(x, taps)                                             # (B, H, W, 6 * Cin), tap-major
    w2 = w.permute(0, 2, 3, 1).reshape(Cout, -1, 1, 1).contiguous()
    a = conv_hip.conv(conv_hip.CONV_1x1, cols, nn.Module(), w2, None, cols.shape[3], Cout, 0)
    b = conv_hip.conv_taps(x, nn.Module(), w, None, taps)
    lim = 1e-5 * float(b.abs().max()) + 1e-4 * b.abs()
    assert bool(((a - b).abs() <= lim).all())
