"""INTEGRATION.md section B, executed: integration/vqvae_hip_stub.py is the binding a maintainer of the reference would add
(ctypes + torch only).  On the build container it is applied to the REAL reference classes (structure / state_dict
contract; no GPU there); on the GPU box it runs numerically on a module with the reference's state_dict keys."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub():
    spec = importlib.util.spec_from_file_location("vqvae_hip_stub", os.path.join(ROOT, "integration", "vqvae_hip_stub.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stub_structs_match_the_header():
    """field order of the ctypes structs == include/vqvae_hip.h (the stub re-declares them independently)"""
    from vqvae_amd import _lib
    stub = _stub()
    assert [f for f, _ in stub.Dims._fields_] == [f for f, _ in _lib.VqvaeDims._fields_]
    assert [f for f, _ in stub.RawWeights._fields_] == [f for f, _ in _lib.VqvaeRawWeights._fields_]
    assert C_sizeof(stub.Weights) == C_sizeof(_lib.VqvaeWeights)
    hdr = open(os.path.join(ROOT, "include", "vqvae_hip.h")).read()
    raw = hdr[hdr.index("typedef struct VqvaeRawWeights"):hdr.index("} VqvaeRawWeights;")]
    pos = [raw.index(f) for f, _ in stub.RawWeights._fields_]
    assert pos == sorted(pos)


def C_sizeof(t):
    import ctypes
    return ctypes.sizeof(t)


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="the reference tree is only in the build container")
def test_install_on_the_real_reference_class():
    """The stub reads the reference's own VQVAE through state_dict(): every key it needs exists with the expected shape,
    install() keeps the constructor / attributes untouched, and the patched forward refuses CPU tensors loudly."""
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    try:
        from models.vqvae import VQVAE
    finally:
        sys.path.remove("/root/reference")
    stub = _stub()
    torch.manual_seed(0)
    m = VQVAE(128, 32, 2, 512, 64, 0.25).eval()
    sd = m.state_dict()
    for f, k in stub.RAW:
        assert k in sd, k
    assert sd["decoder.inverse_conv_stack.0.weight"].shape == (64, 128, 3, 3)          # ConvTranspose2d: (Cin, Cout, kh, kw)
    stub.install(m)
    assert hasattr(m, "hip_repack") and m.vector_quantization.beta == 0.25
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.randn(2, 3, 32, 32))


@pytest.mark.gpu
def test_installed_forward_matches_reference_golden(golden_models):
    """numerics of the installed forward (vqvae_forward_f32 through the stub) against the reference's goldens"""
    from tests.test_model_gpu import build
    stub = _stub()
    name = "kat1"
    m, x = build(name)                                     # same state_dict keys / init bits as the reference's class
    m = m.to("cuda:0")
    plain = torch.nn.Module.__new__(type("RefLike", (torch.nn.Module,), {}))      # a bare module carrying only the parameters
    torch.nn.Module.__init__(plain)
    for attr in ("encoder", "pre_quantization_conv", "vector_quantization", "decoder"):
        setattr(plain, attr, getattr(m, attr))
    stub.install(plain)
    with torch.no_grad():
        loss, x_hat, ppl = plain(x.to("cuda:0"))
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss.item(), golden_models[f"{name}/loss"], rtol=1e-5)
    np.testing.assert_allclose(ppl.item(), golden_models[f"{name}/perplexity"], rtol=1e-5)
    np.testing.assert_allclose(x_hat.cpu().numpy(), golden_models[f"{name}/x_hat"], atol=1e-5, rtol=1e-4)


@pytest.mark.gpu
def test_installed_encode_decode_wire_format(golden_models):
    """the stub's encode / decode_indices (vqvae_encode_f32 / vqvae_decode_f32 through ctypes only): indices equal the installed
    forward's reference-golden indices up to provable near-ties, x_hat of the round trip equals the forward's to fp32 rounding"""
    from tests.test_model_gpu import build
    stub = _stub()
    name = "kat1"
    h, rh, nl, K, D, beta, B, H, W = cases.MODEL_CASES[name]
    m, x = build(name)
    m = m.to("cuda:0")
    plain = torch.nn.Module.__new__(type("RefLike", (torch.nn.Module,), {}))
    torch.nn.Module.__init__(plain)
    for attr in ("encoder", "pre_quantization_conv", "vector_quantization", "decoder"):
        setattr(plain, attr, getattr(m, attr))
    stub.install(plain)
    xd = x.to("cuda:0")
    with torch.no_grad():
        idx = plain.encode(xd)
        x_hat = plain.decode_indices(idx, B, H // 4, W // 4)
        _, x_hat_f, _ = plain(xd)
    torch.cuda.synchronize()
    want = golden_models[f"{name}/idx"].astype(np.int64).reshape(-1)
    assert (idx.view(-1).cpu().numpy() != want).sum() <= max(1, int(1e-4 * want.size))
    np.testing.assert_allclose(x_hat.cpu().numpy(), x_hat_f.cpu().numpy(), atol=1e-6, rtol=1e-5)
    with pytest.raises(IndexError):
        plain.decode_indices(torch.full_like(idx, K), B, H // 4, W // 4)


@pytest.mark.gpu
def test_installed_quantizer_matches_reference_golden(golden_vq):
    stub = _stub()

    class RefLikeQuantizer(torch.nn.Module):               # the attributes models/quantizer.py:20-27 defines
        def __init__(self, n_e, e_dim, beta):
            super().__init__()
            self.n_e, self.e_dim, self.beta = n_e, e_dim, beta
            self.embedding = torch.nn.Embedding(n_e, e_dim)

    name = "k512_d64_c1"
    z, cb, beta = cases.vq_inputs(name)
    vq = RefLikeQuantizer(cb.shape[0], cb.shape[1], beta).to("cuda:0")
    with torch.no_grad():
        vq.embedding.weight.copy_(cb)
        stub.install_quantizer(vq)
        loss, z_q, ppl, onehot, idx = vq(z.to("cuda:0"))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(idx.cpu().numpy().reshape(-1), golden_vq[f"{name}/idx"].astype(np.int64))
    assert cases.sha(z_q) == golden_vq[f"{name}/sha"][2]
    assert onehot.shape == (idx.numel(), cb.shape[0]) and float(onehot.sum()) == idx.numel()
    np.testing.assert_allclose(loss.item(), golden_vq[f"{name}/loss"], rtol=1e-6)
