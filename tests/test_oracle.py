"""CPU tests (-m "not gpu"): pin the oracle.

1. C oracle vs the committed golden vectors produced by the real reference.
2. C oracle's rounding order vs live torch CPU ops (the order probes of SURVEY.md A.1).
3. torch_port (the CPU-baseline restatement) vs golden, and -- where /root/reference
   exists -- bitwise vs the imported reference.
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import c_oracle, torch_port
from tests import cases
from tests.conftest import have_reference

VQ_NAMES = list(cases.VQ_CASES)


@pytest.mark.parametrize("name", VQ_NAMES)
def test_c_oracle_vq_matches_reference_golden(name, golden_vq):
    z, cb, beta = cases.vq_inputs(name)
    sha = golden_vq[f"{name}/sha"]
    assert cases.sha(z) == sha[0] and cases.sha(cb) == sha[1], "input generator drifted"
    out = c_oracle.vq_forward(z.numpy(), cb.numpy(), beta)
    # bit-exact indices (int64 (N,1)) and z_q
    np.testing.assert_array_equal(out["idx"].reshape(-1), golden_vq[f"{name}/idx"].astype(np.int64))
    assert cases.sha(out["idx"]) == sha[3]
    assert cases.sha(out["z_q"]) == sha[2], "z_q not bit-exact"
    # scalars: rtol 1e-6 (fp32 reduction order)
    np.testing.assert_allclose(out["loss"], golden_vq[f"{name}/loss"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(out["perplexity"], golden_vq[f"{name}/perplexity"], rtol=1e-6)
    assert out["hist"].sum() == out["idx"].size


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 24, 31, 32, 33, 40, 48, 63, 64, 96, 100, 128, 200, 256, 511, 512, 513, 600, 1000, 1024, 4096])
def test_row_sqnorm_order_matches_torch(d):
    g = torch.Generator().manual_seed(d)
    for scale in (1.0, 1e-3):
        x = torch.randn(4096, d, generator=g) * scale
        ref = torch.sum(x ** 2, dim=1).numpy()
        got = c_oracle.row_sqnorm(x.numpy())
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), \
            f"sum(x**2) order differs from ATen on this host for D={d}"


@pytest.mark.parametrize("K,D", [(512, 64), (1024, 64), (8192, 128), (100, 32), (64, 256), (96, 48), (50, 7), (33, 5), (40, 200), (77, 255), (64, 1)])
def test_distance_matrix_bitwise_vs_torch(K, D):
    """d = sum(z^2) + sum(e^2) - 2 z@E^T element-for-element (models/quantizer.py:49-51)."""
    g = torch.Generator().manual_seed(K + D)
    N = 256
    z = torch.randn(N, D, generator=g) * 0.07
    cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
    d_ref = (torch.sum(z ** 2, dim=1, keepdim=True) + torch.sum(cb ** 2, dim=1)
             - 2 * torch.matmul(z, cb.t())).numpy()
    out = c_oracle.vq_forward(z.t().reshape(1, D, 1, N).numpy() if False else
                              z.view(N, D, 1, 1).numpy(), cb.numpy(), 0.25, want_dist=True)
    assert np.array_equal(d_ref.view(np.uint32), out["dist"].view(np.uint32))
    assert np.array_equal(out["idx"].reshape(-1), torch.argmin(torch.from_numpy(d_ref), 1).numpy())


@pytest.mark.parametrize("name", list(cases.MODEL_CASES))
def test_torch_port_matches_reference_golden(name, golden_models):
    h, rh, nl, K, D, beta, B, H, W = cases.MODEL_CASES[name]
    sd = torch_port.init_state_dict(h, rh, K, D, n_res_layers=nl)
    keys = list(golden_models[f"{name}/keys"])
    sha = list(golden_models[f"{name}/sha"])
    assert set(sd.keys()) == set(keys)
    for k, s in zip(keys, sha[5:]):
        assert cases.sha(sd[k]) == s, f"weight init drifted from the reference: {k}"
    x = cases.model_inputs(name)
    assert cases.sha(x) == sha[0]
    loss, x_hat, ppl, z_e, z_q, idx = torch_port.forward(sd, x, beta, nl, full=True)
    np.testing.assert_array_equal(idx.numpy().reshape(-1), golden_models[f"{name}/idx"])
    # same ATen ops in the same order -> bitwise on the generating host; tolerance elsewhere
    np.testing.assert_allclose(z_e.numpy(), golden_models[f"{name}/z_e"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(x_hat.numpy(), golden_models[f"{name}/x_hat"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(loss.numpy(), golden_models[f"{name}/loss"], rtol=1e-6)
    np.testing.assert_allclose(ppl.numpy(), golden_models[f"{name}/perplexity"], rtol=1e-6)


@pytest.mark.parametrize("name", list(cases.MODEL_CASES))
def test_c_oracle_model_matches_reference_golden(name, golden_models):
    """Full C restatement (convs correctly rounded, quantizer bit-exact given z_e).
    End-to-end parity tier P1: z_e atol 2e-6; indices exact except
    provable near-ties; x_hat atol 1e-5 + rtol 1e-4 on rows whose index agrees."""
    h, rh, nl, K, D, beta, B, H, W = cases.MODEL_CASES[name]
    if name == "kat1":
        B = 4                      # scalar C convs: keep the CPU suite fast
    sd = torch_port.init_state_dict(h, rh, K, D, n_res_layers=nl)
    x = cases.model_inputs(name)[:B]
    m = c_oracle.Model(sd, beta, nl)
    z_e = m.encode(x.numpy())
    g_ze = golden_models[f"{name}/z_e"][:B]
    np.testing.assert_allclose(z_e, g_ze, atol=2e-6, rtol=0)
    # quantizer boundary (P0): identical z_e bits -> identical indices
    n = B * (H // 4) * (W // 4)
    q = c_oracle.vq_forward(g_ze, m.codebook, beta)
    if B == cases.MODEL_CASES[name][6]:
        np.testing.assert_array_equal(q["idx"].reshape(-1), golden_models[f"{name}/idx"][:n])
    else:
        np.testing.assert_array_equal(q["idx"].reshape(-1), golden_models[f"{name}/idx"][:n])
    x_hat = m.decode(q["z_q"])
    np.testing.assert_allclose(x_hat, golden_models[f"{name}/x_hat"][:B], atol=1e-5, rtol=1e-4)


@pytest.mark.skipif(not have_reference(), reason="reference tree not present (GPU box)")
def test_torch_port_bitwise_vs_imported_reference():
    sys.dont_write_bytecode = True
    ref = os.environ.get("VQVAE_REFERENCE", "/root/reference")
    sys.path.insert(0, ref)
    try:
        import models.quantizer as rq
        from models.vqvae import VQVAE
        rq.device = torch.device("cpu")
        torch.manual_seed(0)
        m = VQVAE(128, 32, 2, 512, 64, 0.25).eval()
        x = torch.randn(8, 3, 32, 32)
        with torch.no_grad():
            loss, x_hat, ppl = m(x)
        sd = {k: v.detach() for k, v in m.state_dict().items()}
        l2, xh2, p2 = torch_port.forward(sd, x.clone(), 0.25, 2)
        assert torch.equal(x_hat, xh2) and torch.equal(loss, l2) and torch.equal(ppl, p2)
    finally:
        sys.path.remove(ref)
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            del sys.modules[k]


def test_residual_quirks():
    """shared weights + relu(x) on the skip."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 5, 6, generator=g)
    w1 = torch.randn(8, 16, 3, 3, generator=g) * 0.1
    w2 = torch.randn(16, 8, 1, 1, generator=g) * 0.1
    ref = torch_port.residual_stack(x.clone(), w1, w2, 3).numpy()
    got = c_oracle.residual_stack(x.numpy(), w1.numpy(), w2.numpy(), 3)
    np.testing.assert_allclose(got, ref, atol=2e-6)


def test_onehot_and_decode_indices():
    z, cb, beta = cases.vq_inputs("k100_d32_ragged")
    out = c_oracle.vq_forward(z.numpy(), cb.numpy(), beta)
    oh = c_oracle.onehot(out["idx"], cb.shape[0])
    assert oh.sum() == out["idx"].size and (oh.argmax(1) == out["idx"].reshape(-1)).all()
    B, D, H, W = z.shape
    zq = c_oracle.decode_indices(out["idx"], cb.numpy(), B, H, W)
    ref = (torch.from_numpy(oh) @ cb).view(B, H, W, D).permute(0, 3, 1, 2).numpy()
    np.testing.assert_array_equal(zq, ref)


def test_hetero_unit_fixture_is_the_oracle():
    """tests/golden/vq_hetero_unit.npz (round 4's regression unit for the quantizer kernels): its stored indices and z_q are
    what the C oracle AND the reference's ATen ops (oracle/torch_port.py) compute for its rows and codebook."""
    import os
    import torch
    from oracle import c_oracle, torch_port
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vq_hetero_unit.npz"))
    z = np.ascontiguousarray(d["z_rows"].reshape(1, 8, 8, 64).transpose(0, 3, 1, 2))
    out = c_oracle.vq_forward(z, d["codebook"], 0.25)
    assert np.array_equal(out["idx"].reshape(-1), d["idx"]) and np.array_equal(out["z_q"].view(np.uint32), d["z_q"].view(np.uint32))
    t = torch_port.quantize(torch.from_numpy(z), torch.from_numpy(d["codebook"]), 0.25)
    assert np.array_equal(t[4].numpy().reshape(-1), d["idx"]) and np.array_equal(t[1].numpy().view(np.uint32), d["z_q"].view(np.uint32))


# ---- round 6: really trained checkpoints (tests/golden/<name>_state.npz + trained_cases.npz, generated by oracle/gen_golden.py
# from the UNMODIFIED reference loaded with the checkpoint) ------------------------------------------------------------------------
@pytest.fixture(scope="module")
def golden_trained():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_cases.npz"))


@pytest.mark.parametrize("name", list(cases.TRAINED_CASES))
def test_oracles_on_a_trained_checkpoint(name, golden_trained):
    """The C oracle's quantizer on the reference's z_e bits: indices and z_q bit-exact on a checkpoint whose codebook and z_e are
    what TRAINING made them (max|z_e| ~ 6, 57 / 125 codes in use) -- not the U(+-1/K) init every other golden rests on; torch_port
    on the same weights and images reproduces the reference's whole forward."""
    from tests import synthdata
    h, rh, nl, K, D, beta, B, seed = cases.TRAINED_CASES[name]
    sd = cases.trained_state(name)
    assert len(sd) == 23
    x = torch.from_numpy(golden_trained[f"{name}/x"])          # (committed: the generator's transcendental ops round per host ISA)
    sha = golden_trained[f"{name}/sha"]
    assert cases.sha(x) == sha[0]
    if have_reference():                                         # the build container: the generator still makes these very images
        assert cases.sha(synthdata.normalised(B, seed)) == sha[0], "synthetic image generator drifted"
    g_ze = golden_trained[f"{name}/z_e"]
    out = c_oracle.vq_forward(g_ze, sd["vector_quantization.embedding.weight"].numpy(), beta)
    np.testing.assert_array_equal(out["idx"].reshape(-1), golden_trained[f"{name}/idx"].astype(np.int64))
    assert cases.sha(out["z_q"]) == sha[2], "z_q not bit-exact"
    np.testing.assert_allclose(out["loss"], golden_trained[f"{name}/loss"], rtol=1e-6)
    np.testing.assert_allclose(out["perplexity"], golden_trained[f"{name}/perplexity"], rtol=1e-6)
    loss, x_hat, ppl, z_e, z_q, idx = torch_port.forward(sd, x, beta, nl, full=True)
    np.testing.assert_array_equal(idx.numpy().reshape(-1), golden_trained[f"{name}/idx"])
    cmax = np.abs(g_ze).max()
    np.testing.assert_allclose(z_e.numpy(), g_ze, atol=2e-6 * cmax, rtol=0)
    np.testing.assert_allclose(x_hat.numpy(), golden_trained[f"{name}/x_hat"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(loss.numpy(), golden_trained[f"{name}/loss"], rtol=1e-6)
    # the C oracle's own convs (plain k-ordered loops, not oneDNN's order): whole model within the fp32 tolerance
    ref = c_oracle.Model(sd, beta, nl).forward(x.numpy())
    np.testing.assert_allclose(ref["z_e"], g_ze, atol=2e-6 * cmax, rtol=0)


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("name", list(cases.TRAINED_CASES))
def test_trained_goldens_reproduce_from_the_unmodified_reference(name, golden_trained):
    from tests import synthdata
    sys.path.insert(0, os.environ.get("VQVAE_REFERENCE", "/root/reference"))
    sys.dont_write_bytecode = True
    import models.quantizer as ref_q
    from models.vqvae import VQVAE
    ref_q.device = torch.device("cpu")
    h, rh, nl, K, D, beta, B, seed = cases.TRAINED_CASES[name]
    m = VQVAE(h, rh, nl, K, D, beta).eval()
    m.load_state_dict(cases.trained_state(name), strict=True)
    with torch.no_grad():
        loss, x_hat, ppl = m(torch.from_numpy(golden_trained[f"{name}/x"]))
    assert cases.sha(x_hat) == golden_trained[f"{name}/sha"][3]
    assert loss.item() == golden_trained[f"{name}/loss"].item()


@pytest.mark.parametrize("name", VQ_NAMES)
def test_fast_index_oracle_equals_the_scalar_one(name, golden_vq):
    """Round 6: vqo_vq_indices_rows (eight codes per AVX2 register, host threads over row slabs -- the checker of the full-size
    configs) against the reference's goldens: every shape incl. ties, NaN / Inf rows, K not a multiple of 32, D = 7 ... 256."""
    z, cb, beta = cases.vq_inputs(name)
    rows = z.permute(0, 2, 3, 1).reshape(-1, z.shape[1]).contiguous().numpy()
    for threads, slab in ((1, 8192), (4, 7)):
        got = c_oracle.vq_indices_rows(rows, cb.numpy(), threads=threads, slab=slab)
        np.testing.assert_array_equal(got, golden_vq[f"{name}/idx"].astype(np.int64))


def test_fast_index_oracle_on_trained_rows_and_random_shapes(golden_trained):
    for name in cases.TRAINED_CASES:
        g_ze = golden_trained[f"{name}/z_e"]
        rows = np.ascontiguousarray(np.transpose(g_ze, (0, 2, 3, 1)).reshape(-1, g_ze.shape[1]))
        cb = cases.trained_state(name)["vector_quantization.embedding.weight"].numpy()
        np.testing.assert_array_equal(c_oracle.vq_indices_rows(rows, cb), golden_trained[f"{name}/idx"].astype(np.int64))
    g = torch.Generator().manual_seed(5)
    for K, D, N in ((1024, 64, 3000), (8192, 128, 300), (33, 48, 100), (1, 64, 10)):
        z = torch.randn(N, D, generator=g) * 0.07
        cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
        want = c_oracle.vq_forward(z.view(N, D, 1, 1).numpy(), cb.numpy(), 0.25)["idx"].reshape(-1)
        np.testing.assert_array_equal(c_oracle.vq_indices_rows(z.numpy(), cb.numpy()), want)
