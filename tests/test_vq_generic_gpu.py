"""GPU parity tests (-m gpu) of the quantizer for embedding widths outside {32, 64, 128, 256} (round 5: vq_generic_kernel, the
exact-fp32 vector path; main.py:21 leaves --embedding_dim free and the reference just runs).  Same contract as the matrix-core
kernels: indices and z_q bit-exact against the oracle / the reference's golden outputs, loss / perplexity rtol 1e-6.  The golden
cases of such widths (tests/cases.py: k96_d48_ragged, k300_d48_init, k50_d7, k40_d200, ties_d48, nonfinite_d72) run in
tests/test_vq_gpu.py::test_vq_matches_reference_golden with every other case."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("mode", [0, 1])
def test_row_sqnorm_in_atens_order_bitwise_vs_torch(mode):
    """||x||^2 per row is the one place where the ORDER of a reduction reaches the indices (quantizer.py:49-50: torch.sum(z ** 2, dim=1),
    a cascade sum on the CPU the reference's bits come from).  Both device forms -- one thread per row (the codebook term), the
    workgroup-cooperative tile form (the row term) -- against live torch, every width class: below one vector, ragged tails, the
    4-way ILP groups, and the cascade levels from 512 on."""
    from vqvae_amd import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(5)
    for D in list(range(1, 41)) + [47, 48, 63, 64, 65, 96, 100, 127, 200, 255, 300, 511, 512, 513, 600, 768, 1000, 1023, 1024]:
        rows = 37
        x = torch.randn(rows, D, generator=g) * torch.logspace(-3, 3, rows).unsqueeze(1)
        want = torch.sum(x ** 2, dim=1).numpy()
        xd, out = x.to(_dev()), torch.empty(rows, device=_dev())
        _lib.check(L.vqvae_debug_row_sqnorm_f32(xd.data_ptr(), rows, D, mode, out.data_ptr(), None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"D={D} mode={mode}"
    assert L.vqvae_debug_row_sqnorm_f32(xd.data_ptr(), 4, 1025, mode, out.data_ptr(), None) == _lib.ERR_UNSUPPORTED


@pytest.mark.parametrize("K,D,B,H,W,scale", [
    (512, 48, 16, 8, 8, 0.066),      # the width VERDICT r4 names, reference-initialised codebook
    (96, 48, 3, 5, 7, 1.0),          # ragged: 105 rows
    (100, 20, 4, 6, 6, 1.0),
    (33, 7, 2, 3, 5, 1.0),           # no 16-byte pieces
    (64, 1, 2, 4, 4, 1.0),           # one channel
    (300, 100, 2, 7, 9, 1.0),
    (40, 200, 1, 4, 5, 1.0),         # a ragged vector tail
    (17, 255, 1, 3, 3, 1.0),         # the widest odd width
    (1, 48, 2, 4, 4, 1.0),           # single code
    (4096, 24, 1, 8, 8, 0.066),      # many codes per thread
])
@pytest.mark.parametrize("vector_units", [False, True], ids=["mfma_fp32", "vector_units"])
def test_vq_generic_matches_oracle_fresh(K, D, B, H, W, scale, vector_units):
    """Both implementations for these widths: round 6's matrix-core kernel (vq_anyd_kernel: rows and codes zero-padded to a multiple
    of eight channels, v_mfma_f32_32x32x2_f32 = the reference's fmaf chain; the default) and round 5's per-thread chains
    (vq_generic_kernel, behind VQVAE_VQ_BF16_FILTER)."""
    from oracle import c_oracle
    from vqvae_amd import _lib, functional as F
    assert _lib.vq_kernel_name(K, D) == "vq_anyd_kernel" and _lib.vq_kernel_name(K, D, 0x1 | 0x8) == "vq_generic_kernel"
    g = torch.Generator().manual_seed(K * 7 + D + B)
    cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K) if scale < 1 else torch.randn(K, D, generator=g)
    z = torch.randn(B, D, H, W, generator=g) * scale
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    for rowmajor in (False, True):
        zd = z.to(_dev())
        if rowmajor:
            zd = zd.permute(0, 2, 3, 1).contiguous()
        loss, zq, ppl, idx, hist = F.vq_forward(zd, cb.to(_dev()), 0.25, rowmajor=rowmajor, bf16_filter=vector_units)
        torch.cuda.synchronize()
        if rowmajor:
            zq = zq.permute(0, 3, 1, 2).contiguous()
        np.testing.assert_array_equal(idx.cpu().numpy(), ref["idx"])
        assert np.array_equal(zq.cpu().numpy().view(np.uint32), ref["z_q"].view(np.uint32))
        np.testing.assert_allclose(loss.item(), ref["loss"], rtol=1e-6)
        np.testing.assert_allclose(ppl.item(), ref["perplexity"], rtol=1e-6)
        np.testing.assert_array_equal(hist.cpu().numpy(), ref["hist"])
        _, none_zq, _, idx2, _ = F.vq_forward(zd, cb.to(_dev()), 0.25, rowmajor=rowmajor, want_zq=False, bf16_filter=vector_units)
        assert none_zq is None and torch.equal(idx2, idx)


@pytest.mark.parametrize("K,D", [(512, 48), (100, 7), (300, 200), (64, 72), (9000, 40)])
def test_vq_anyd_special_values_and_prepared_codebook(K, D):
    """The matrix-core kernel's corners: NaN / Inf / overflowing rows (||z||^2 not < 1e38: vector-unit chains inside the kernel, torch.argmin's
    NaN rule), an Inf in the codebook (every row takes that path), K beyond the LDS histogram, ragged last tile, a prepared codebook image
    reused across calls -- against the oracle, and the two implementations against each other."""
    from oracle import c_oracle
    from vqvae_amd import functional as F
    g = torch.Generator().manual_seed(K + D)
    cb = torch.randn(K, D, generator=g)
    n = 64 * 5 + 13
    zr = torch.randn(n, D, generator=g)
    zr[3, D // 2] = float("nan")
    zr[10, 0] = float("inf")
    zr[11, D - 1] = float("-inf")
    zr[12, :] = 3.0e19                                 # zz overflows to +inf
    zr[70, 1 % D] = 1.0e30
    zr[n - 1, 0] = float("nan")
    z = zr.view(n, 1, 1, D).permute(0, 3, 1, 2).contiguous()
    for bad_cb in (False, True):
        if bad_cb:
            cb = cb.clone()
            cb[K // 2, D // 3] = float("inf")
        ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
        zd, cbd = z.to(_dev()), cb.to(_dev())
        ws = F.vq_workspace(K, D, _dev())
        a = F.vq_forward(zd, cbd, 0.25, workspace=ws)
        b = F.vq_forward(zd, cbd, 0.25, workspace=ws, prepared=True)           # the same images again
        c = F.vq_forward(zd, cbd, 0.25, bf16_filter=True)                       # round 5's kernel
        torch.cuda.synchronize()
        for out in (a, b, c):
            np.testing.assert_array_equal(out[3].cpu().numpy(), ref["idx"])
            got, want = out[1].cpu().numpy(), ref["z_q"]
            assert np.array_equal(np.isnan(got), np.isnan(want))
            m = ~np.isnan(want)
            assert np.array_equal(got[m].view(np.uint32), want[m].view(np.uint32))
            np.testing.assert_array_equal(out[4].cpu().numpy(), ref["hist"])


def test_vq_generic_near_ties_and_duplicates_take_the_first_index():
    """Exact duplicates and 1e-7-close codes at a width the screens never see: every distance is exact fp32 here, so the reference's
    first-index rule must hold by construction; 32 768 rows so that every workgroup and a ragged last tile take part."""
    from oracle import c_oracle
    from vqvae_amd import functional as F
    g = torch.Generator().manual_seed(11)
    K, D = 200, 48
    base = torch.randn(10, D, generator=g)
    cb = base[torch.randint(0, 10, (K,), generator=g)].clone()
    cb[50:] += torch.randn(K - 50, D, generator=g) * 1e-7          # codes 0..49: exact duplicates of the ten prototypes
    n = 32768 + 5
    zr = base[torch.randint(0, 10, (n,), generator=g)] + torch.randn(n, D, generator=g) * 1e-4
    zr[::9] = 0.0
    z = zr.view(n, 1, 1, D).permute(0, 3, 1, 2).contiguous()
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    loss, zq, ppl, idx, hist = F.vq_forward(z.to(_dev()), cb.to(_dev()), 0.25)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(idx.cpu().numpy(), ref["idx"])
    assert np.array_equal(zq.cpu().numpy().view(np.uint32), ref["z_q"].view(np.uint32))
    np.testing.assert_allclose(loss.item(), ref["loss"], rtol=1e-6)


def test_module_boundary_with_an_unusual_width():
    """VectorQuantizer(n_e, e_dim = 48, beta).forward -- the reference's five outputs (models/quantizer.py:76), one-hot included."""
    from oracle import torch_port
    from vqvae_amd.modules import VectorQuantizer
    torch.manual_seed(3)
    vq = VectorQuantizer(96, 48, 0.25).to(_dev())
    z = torch.randn(2, 48, 6, 5)
    with torch.no_grad():
        loss, z_q, ppl, onehot, idx = vq(z.to(_dev()))
        want = torch_port.quantize(z, vq.embedding.weight.detach().cpu(), 0.25)
    torch.cuda.synchronize()
    assert torch.equal(idx.cpu(), want[4]) and torch.equal(z_q.cpu(), want[1]) and torch.equal(onehot.cpu(), want[3])
    np.testing.assert_allclose(loss.item(), want[0].item(), rtol=1e-6)
    np.testing.assert_allclose(ppl.item(), want[2].item(), rtol=1e-6)
