"""GPU parity tests (-m gpu) for the fused VectorQuantizer kernel, through the C ABI.

Tier P0: for identical z_e bits, min_encoding_indices and z_q are
BIT-EXACT against (a) the committed golden vectors produced by the real reference and
(b) the C oracle on fresh seeded inputs; loss / perplexity rtol 1e-6.
"""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _run(z, cb, beta, rowmajor=False, want_zq=True, exact=False, bf16_filter=False, form=0):
    from vqvae_amd import functional as F
    zd = z.to(_dev())
    if rowmajor:
        zd = zd.permute(0, 2, 3, 1).contiguous()
    loss, zq, ppl, idx, hist = F.vq_forward(zd, cb.to(_dev()), beta, rowmajor=rowmajor, want_zq=want_zq,
                                            exact_sweep=exact, bf16_filter=bf16_filter, form=form)
    torch.cuda.synchronize()
    if zq is not None and rowmajor:
        zq = zq.permute(0, 3, 1, 2).contiguous()
    return loss.cpu().numpy(), None if zq is None else zq.cpu().numpy(), ppl.cpu().numpy(), \
        idx.cpu().numpy(), hist.cpu().numpy()


@pytest.mark.parametrize("kernel", ["default", "units64_8waves", "units32_16waves", "bf16_filter", "exact"])
@pytest.mark.parametrize("rowmajor", [False, True])
@pytest.mark.parametrize("name", list(cases.VQ_CASES))
def test_vq_matches_reference_golden(name, rowmajor, kernel, golden_vq):
    """All three kernels -- the single-sweep fp16-screened one (default for row-major D=64 rows), round 1's
    two-sweep bf16 filter (default for NCHW D=64 rows) and the exhaustive fp32-MFMA sweep -- must reproduce
    the reference bit for bit."""
    z, cb, beta = cases.vq_inputs(name)
    loss, zq, ppl, idx, hist = _run(z, cb, beta, rowmajor, exact=kernel == "exact", bf16_filter=kernel == "bf16_filter", form={"units64_8waves": 8, "units32_16waves": 16}.get(kernel, 0))
    assert idx.shape == (z.shape[0] * z.shape[2] * z.shape[3], 1) and idx.dtype == np.int64
    np.testing.assert_array_equal(idx.reshape(-1), golden_vq[f"{name}/idx"].astype(np.int64))
    sha = golden_vq[f"{name}/sha"]
    if name.startswith("nonfinite"):
        # NaN payload/sign is ISA-specific (x86 default NaN is 0xFFC00000, gfx950's 0x7FC00000):
        # NaN positions must agree, every non-NaN element must be bit-identical
        g = golden_vq[f"{name}/z_q"]
        assert np.array_equal(np.isnan(zq), np.isnan(g))
        m = ~np.isnan(g)
        assert np.array_equal(zq[m].view(np.uint32), g[m].view(np.uint32))
    else:
        assert cases.sha(zq) == sha[2], "z_q not bit-exact vs the reference"
    np.testing.assert_allclose(loss, golden_vq[f"{name}/loss"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(ppl, golden_vq[f"{name}/perplexity"], rtol=1e-6)
    np.testing.assert_array_equal(hist, np.bincount(idx.reshape(-1), minlength=cb.shape[0]))


@pytest.mark.parametrize("K,D,B,H,W,scale", [
    (512, 64, 64, 8, 8, 0.066),      # 4096 rows, benchmark codebook
    (512, 64, 33, 8, 8, 1.0),        # ragged row blocks
    (1024, 64, 2, 56, 56, 0.066),    # config-4 shape: two LDS chunks, 3136-row images
    (8192, 128, 1, 32, 32, 0.066),   # config-5 shape: 37 chunks
    (1000, 128, 3, 7, 9, 1.0),       # streamed kernel: K % 32 != 0, 189 rows (one ragged block)
    (700, 64, 5, 9, 13, 1.0),        # streamed kernel, D = 64: 585 rows
    (512, 32, 7, 9, 5, 1.0),
    (300, 256, 2, 6, 6, 1.0),
    (1, 64, 2, 4, 4, 1.0),           # single code
    # round 4: the stream-tracker kernel on NCHW input (maps whose pixel count is a multiple of 64: a unit = 64 positions of
    # one image) -- several units per image, a 3136-pixel map (49 units), K % 32 != 0
    (512, 64, 3, 16, 16, 0.066),
    (500, 64, 40, 8, 16, 1.0),
    (256, 64, 2, 56, 56, 0.066),
])
def test_vq_matches_oracle_fresh(K, D, B, H, W, scale):
    from oracle import c_oracle
    g = torch.Generator().manual_seed(K * 7 + D + B)
    cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K) if scale < 1 else torch.randn(K, D, generator=g)
    z = torch.randn(B, D, H, W, generator=g) * scale
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    for rowmajor, exact, bf in ((False, False, False), (True, False, False), (True, False, True), (False, True, False),
                                (True, True, False)):
        loss, zq, ppl, idx, hist = _run(z, cb, 0.25, rowmajor, exact=exact, bf16_filter=bf)
        np.testing.assert_array_equal(idx, ref["idx"])
        assert np.array_equal(zq.view(np.uint32), ref["z_q"].view(np.uint32))
        np.testing.assert_allclose(loss, ref["loss"], rtol=1e-6)
        np.testing.assert_allclose(ppl, ref["perplexity"], rtol=1e-6)
        np.testing.assert_array_equal(hist, ref["hist"])


def test_vq_filter_adversarial_near_ties():
    """Stress the screen's bound: codes that differ from each other by ~1 ulp-level perturbations, rows
    sitting almost exactly between codes, huge dynamic range, and > CAP near-duplicates per row
    (candidate-list overflow -> scalar path).  The filter kernel must still equal the oracle bit for bit."""
    from oracle import c_oracle
    g = torch.Generator().manual_seed(77)
    K, D = 512, 64
    base = torch.randn(8, D, generator=g)
    cb = base[torch.randint(0, 8, (K,), generator=g)].clone()
    cb += torch.randn(K, D, generator=g) * 1e-6            # 64 near-duplicates of each of 8 prototypes
    cb[100:140] = cb[100]                                   # 40 exact duplicates
    cb[300:] *= torch.logspace(-3, 3, K - 300).unsqueeze(1)
    z = torch.empty(6, D, 8, 8)
    zr = base[torch.randint(0, 8, (384,), generator=g)] + torch.randn(384, D, generator=g) * 1e-3
    zr[::7] = 0.5 * (cb[5] + cb[200])                       # midpoints
    zr[3::11] = cb[100]
    zr[5::13] *= 1e3
    zr[6::17] *= 1e-4
    z = zr.view(6, 8, 8, D).permute(0, 3, 1, 2).contiguous()
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    for rowmajor, bf in ((False, False), (True, False), (True, True)):
        loss, zq, ppl, idx, hist = _run(z, cb, 0.25, rowmajor, bf16_filter=bf)
        np.testing.assert_array_equal(idx, ref["idx"])
        assert np.array_equal(zq.view(np.uint32), ref["z_q"].view(np.uint32))
        np.testing.assert_allclose(loss, ref["loss"], rtol=1e-6)


@pytest.mark.parametrize("p", [8, 11], ids=["bf16mid", "fp16mid"])
@pytest.mark.parametrize("seed", range(6))
def test_vq_aligned_rounding_adversarial(p, seed):
    """VERDICT round 1 counter-example, generalised (tests/adversarial.py): every channel of z and of a code
    pair sits just above / below a bf16 (p=8) or fp16 (p=11) midpoint with the signs chosen so that all 64
    roundings move the screened dot product the same way, while the true fp32 margin has the opposite sign.
    A screen whose threshold is too tight by 2x returns the wrong index on 3/4 of these rows
    (tests/test_adversarial.py proves that on the CPU); every kernel must equal the oracle bit for bit."""
    from oracle import c_oracle
    from tests import adversarial as A
    zr, cb, _ = A.make_problem(p, seed=seed)
    n, d = zr.shape
    z = torch.from_numpy(np.ascontiguousarray(zr.reshape(n // 64, 8, 8, d).transpose(0, 3, 1, 2)))
    cbt = torch.from_numpy(cb)
    ref = c_oracle.vq_forward(z.numpy(), cb, 0.25)
    for rowmajor, exact, bf in ((True, False, False), (True, False, True), (False, False, False), (True, True, False)):
        loss, zq, ppl, idx, hist = _run(z, cbt, 0.25, rowmajor, exact=exact, bf16_filter=bf)
        np.testing.assert_array_equal(idx, ref["idx"])
        assert np.array_equal(zq.view(np.uint32), ref["z_q"].view(np.uint32))
        np.testing.assert_allclose(loss, ref["loss"], rtol=1e-6)


@pytest.mark.parametrize("shrink", [1.0, 0.5, -0.25])
def test_vq_near_ties_within_one_lane_half(shrink):
    """Two near-tied codes that the single-sweep kernel tracks in the SAME lane (same half of a 32-code tile
    group, different tiles), with positive, negative and mixed-sign screen scores.  Their keys share every
    upper bit, so the order of the low (code) bits decides -- the end-of-tile fix-up must not reorder them
    (a round-2 bug: with negative scores the later tile's key overtook the earlier one, med3 then duplicated
    it and the true argmin was never refined)."""
    from oracle import c_oracle
    g = torch.Generator().manual_seed(5)
    K, D = 512, 64
    cb = torch.randn(K, D, generator=g)
    rows = []
    for i in range(512):
        ta, tb = torch.randint(0, 16, (2,), generator=g).tolist()
        half = int(torch.randint(0, 2, (1,), generator=g))
        ia = [j for j in range(32) if ((j >> 2) & 1) == half]
        ka = ta * 32 + ia[int(torch.randint(0, 16, (1,), generator=g))]
        kb = tb * 32 + ia[int(torch.randint(0, 16, (1,), generator=g))]
        mid = 0.5 * (cb[ka] + cb[kb])
        rows.append(shrink * mid + torch.randn(D, generator=g) * (1e-4 if i % 2 else 1e-6))
    z = torch.stack(rows).view(8, 8, 8, D).permute(0, 3, 1, 2).contiguous()
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    for rowmajor, bf in ((True, False), (True, True), (False, False)):
        loss, zq, ppl, idx, hist = _run(z, cb, 0.25, rowmajor, bf16_filter=bf)
        np.testing.assert_array_equal(idx, ref["idx"])
        assert np.array_equal(zq.view(np.uint32), ref["z_q"].view(np.uint32))


ALL_FORMS = (("track rows", dict(rowmajor=True)), ("track rows 8 waves", dict(rowmajor=True, form=8)), ("track rows 16 waves", dict(rowmajor=True, form=16)),
             ("track nchw", dict(rowmajor=False)), ("bf16 filter", dict(rowmajor=True, bf16_filter=True)), ("bf16 filter nchw", dict(rowmajor=False, bf16_filter=True)),
             ("exact", dict(rowmajor=True, exact=True)))


def test_vq_heterogeneous_unit_regression():
    """Round 4: the 64-row unit of trained-like z_e (channels six decades apart) on which round 3's stream tracker returned a
    wrong index -- its row norm |z^|^2, which the screen's bound DELTA is built on, came from four fdot2 builtins that hipcc
    compiled to four reads of the SAME register (csrc/common.h, sqsum8_f16): 105 instead of 13 667 for row 21, DELTA a tenth of
    what it must be, the true argmin screened out.  Fixture: tests/golden/vq_hetero_unit.npz (rows, codebook, the C oracle's
    indices and z_q; tests/test_oracle.py checks it against the oracle on the CPU).  Every kernel form, bit for bit."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vq_hetero_unit.npz"))
    cb = torch.from_numpy(d["codebook"])
    rows = torch.from_numpy(d["z_rows"])
    for reps in (1, 5):                                    # one unit; five (a wave with several units, other rows' tasks around)
        z = rows.repeat(reps, 1).reshape(reps, 8, 8, 64).permute(0, 3, 1, 2).contiguous()
        for name, kw in ALL_FORMS:
            loss, zq, ppl, idx, hist = _run(z, cb, 0.25, **kw)
            assert np.array_equal(idx.reshape(reps, 64), np.tile(d["idx"], (reps, 1))), f"{name}: indices"
            assert np.array_equal(zq.view(np.uint32), np.tile(d["z_q"], (reps, 1, 1, 1)).view(np.uint32)), f"{name}: z_q"


@pytest.mark.parametrize("seed", range(4))
def test_vq_row_energy_outside_the_first_channel_pair_of_every_chunk(seed):
    """The same defect, synthetically: rows whose channels 8 c and 8 c + 1 are ZERO (the miscompiled row norm was four times
    the energy of exactly those channels, i.e. 0 here -> DELTA ~ 0 -> the screen's own argmax won) against a codebook of
    near-duplicate pairs, where fp16 rounding decides the screen's order about half the time.  Against the C oracle."""
    from oracle import c_oracle
    g = torch.Generator().manual_seed(900 + seed)
    K, D = 512, 64
    cb = torch.randn(K, D, generator=g)
    cb[1::2] = cb[0::2] + 2e-4 * torch.randn(K // 2, D, generator=g)            # 256 near-duplicate pairs
    sel = torch.randint(0, K // 2, (1024,), generator=g) * 2
    zr = 0.5 * (cb[sel] + cb[sel + 1]) + 1e-4 * torch.randn(1024, D, generator=g)
    zr.view(1024, 8, 8)[:, :, :2] = 0.0                                         # channels 8 c, 8 c + 1
    z = zr.view(16, 8, 8, D).permute(0, 3, 1, 2).contiguous()
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    for name, kw in ALL_FORMS:
        loss, zq, ppl, idx, hist = _run(z, cb, 0.25, **kw)
        np.testing.assert_array_equal(idx, ref["idx"], err_msg=name)
        assert np.array_equal(zq.view(np.uint32), ref["z_q"].view(np.uint32)), name


def _oracle_vq_chunked(z, cb, beta, n_chunks=64, workers=32):
    """C oracle over image chunks in a thread pool (rows are independent; ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle
    zs = np.array_split(z.numpy(), n_chunks, axis=0)
    cbn = cb.numpy()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        outs = list(ex.map(lambda a: c_oracle.vq_forward(np.ascontiguousarray(a), cbn, beta), zs))
    return np.concatenate([o["idx"] for o in outs]), np.concatenate([o["z_q"] for o in outs])


@pytest.mark.parametrize("B,H,W", [(4096, 8, 8), (2100, 7, 9)], ids=["config3_262144rows", "ragged_132300rows"])
def test_vq_headline_size_bit_exact_vs_oracle(B, H, W):
    """BASELINE config-3 size against the ORACLE (not against the kernels themselves): 262 144 rows of the benchmark
    distribution, and a ragged row count above 131 072 (rows % 64 != 0, several pairs per wave) -- indices and z_q
    bit for bit from the default kernel, its producer/consumer form and round 1's bf16 filter."""
    g = torch.Generator().manual_seed(2024 + B)
    K, D = 512, 64
    cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
    z = torch.randn(B, D, H, W, generator=g) * 0.066
    ref_idx, ref_zq = _oracle_vq_chunked(z, cb, 0.25)
    for kw in ({}, {"form": 8}, {"form": 16}, {"bf16_filter": True}):
        loss, zq, ppl, idx, hist = _run(z, cb, 0.25, True, **kw)
        np.testing.assert_array_equal(idx, ref_idx)
        assert np.array_equal(zq.view(np.uint32), ref_zq.view(np.uint32))
        np.testing.assert_array_equal(hist, np.bincount(ref_idx.reshape(-1), minlength=K))
    loss, zq, ppl, idx, hist = _run(z, cb, 0.25, False)          # NCHW boundary layout
    np.testing.assert_array_equal(idx, ref_idx)
    assert np.array_equal(zq.view(np.uint32), ref_zq.view(np.uint32))


def test_vq_pooled_tail_units_and_counter_reset_vs_oracle():
    """From four units per wave on (589 824 rows on 256 CUs: 8-wave form and the forced 16-wave form alike) the last quarter of the
    units comes from per-group counters in the workspace (vq_track.hip, pool_pct) that the last workgroup of a group puts back to
    zero: three launches on ONE workspace must each reproduce the oracle bit for bit, row-major and NCHW."""
    from vqvae_amd import functional as F
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    K, D, B = 512, 64, 9216
    cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
    z = torch.randn(B, D, 8, 8, generator=g) * 0.066
    ref_idx, ref_zq = _oracle_vq_chunked(z, cb, 0.25)
    cbd, zn = cb.to(dev), z.to(dev)
    zr = zn.permute(0, 2, 3, 1).contiguous()
    hist_ref = np.bincount(ref_idx.reshape(-1), minlength=K)
    for kw, zin, rm in (({}, zr, True), ({"form": 16}, zr, True), ({}, zn, False)):
        ws = F.vq_workspace(K, D, dev)
        for rep in range(3):
            loss, zq, ppl, idx, hist = F.vq_forward(zin, cbd, 0.25, rowmajor=rm, workspace=ws, prepared=rep > 0, **kw)
            if rm:
                zq = zq.permute(0, 3, 1, 2).contiguous()
            np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx, err_msg=f"{kw} nchw={not rm} launch {rep}")
            assert np.array_equal(zq.cpu().numpy().view(np.uint32), ref_zq.view(np.uint32)), (kw, rm, rep)
            np.testing.assert_array_equal(hist.cpu().numpy(), hist_ref)


@pytest.mark.parametrize("B", [1, 37, 1024, 1025], ids=lambda b: f"{b * 64}rows")
def test_vq_few_rows_take_the_eight_wave_32_row_form_bit_exact(B):
    """Up to 8 x CUs x 32 rows (BASELINE config 2's 65 536 on 256 CUs) the default launch is 32-row units on eight waves per CU;
    one image, a ragged count, the last size of the rule and the first one past it against the oracle."""
    g = torch.Generator().manual_seed(300 + B)
    K, D = 512, 64
    cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
    z = torch.randn(B, D, 8, 8, generator=g) * 0.066
    ref_idx, ref_zq = _oracle_vq_chunked(z, cb, 0.25, n_chunks=min(B, 64))
    loss, zq, ppl, idx, hist = _run(z, cb, 0.25, True)
    np.testing.assert_array_equal(idx, ref_idx)
    assert np.array_equal(zq.view(np.uint32), ref_zq.view(np.uint32))
    np.testing.assert_array_equal(hist, np.bincount(ref_idx.reshape(-1), minlength=K))


@pytest.mark.parametrize("B,H,W", [(5, 4, 8), (300, 8, 12), (1500, 12, 8), (1024, 8, 8), (1030, 8, 8)],
                         ids=["hw32", "hw96", "hw96_144000rows", "hw64_last_of_the_rule", "hw64_first_past_it"])
def test_vq_nchw_units_of_32_positions_bit_exact(B, H, W):
    """The module's own NCHW layout on the stream-tracker kernel with units of 32 positions of one image (round 4, second
    session): maps whose pixel count is a multiple of 32 but not of 64 at any row count, and 8x8 maps while the rule for few rows
    holds (the 64-position form beyond it) -- against the oracle, bit for bit."""
    from vqvae_amd import _lib
    assert _lib.vq_kernel_name(512, 64, 0x0) == "vq_track_kernel_d64"
    g = torch.Generator().manual_seed(900 + B + H)
    K, D = 512, 64
    cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
    z = torch.randn(B, D, H, W, generator=g) * 0.066
    ref_idx, ref_zq = _oracle_vq_chunked(z, cb, 0.25, n_chunks=min(B, 64))
    loss, zq, ppl, idx, hist = _run(z, cb, 0.25, False)
    np.testing.assert_array_equal(idx, ref_idx)
    assert np.array_equal(zq.view(np.uint32), ref_zq.view(np.uint32))
    np.testing.assert_array_equal(hist, np.bincount(ref_idx.reshape(-1), minlength=K))


@pytest.mark.parametrize("K,B,H,W", [(1024, 2, 56, 56), (1024, 37, 8, 8), (640, 9, 8, 12), (1000, 3, 16, 16)],
                         ids=["config4_56x56_k1024", "k1024_8x8_ragged", "k640_hw96", "k1000_padding_codes"])
def test_vq_nchw_large_codebooks_on_the_four_wave_form_bit_exact(K, B, H, W):
    """Round 5: the module's NCHW layout with codebooks whose fp16 image leaves 4 KiB of LDS per wave (K up to 1024,
    BASELINE config 4's codebook) runs vq_track_kernel_d64<4, true, 1> -- the 32 x 64 fp32 block is turned around in two halves of 32
    channels through a 32 x 32 tile, in and out -- instead of round 1's two-sweep filter kernel; against the oracle, bit for bit
    (indices, z_q, histogram), incl. a K that is not a multiple of 32 (padding codes) and a ragged number of units."""
    from vqvae_amd import _lib
    assert _lib.vq_kernel_name(K, 64, 0x0) == "vq_track_kernel_d64"
    assert _lib.vq_launch_form(B * H * W, K, 64, H * W, 0x0)[:2] == (4, 32)
    g = torch.Generator().manual_seed(1300 + K + B)
    D = 64
    cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
    z = torch.randn(B, D, H, W, generator=g) * 0.066
    ref_idx, ref_zq = _oracle_vq_chunked(z, cb, 0.25, n_chunks=min(B, 8))
    loss, zq, ppl, idx, hist = _run(z, cb, 0.25, False)
    np.testing.assert_array_equal(idx, ref_idx)
    assert np.array_equal(zq.view(np.uint32), ref_zq.view(np.uint32))
    np.testing.assert_array_equal(hist, np.bincount(ref_idx.reshape(-1), minlength=K))


def test_vq_stream_kernel_is_the_default_for_large_codebooks():
    from vqvae_amd import _lib
    assert _lib.vq_kernel_name(512, 64) == "vq_track_kernel_d64"
    assert _lib.vq_kernel_name(512, 64, 0x1 | 0x10) == "unsupported"              # round 2's tracker kernel and its flags: removed in round 4
    for K, D in ((640, 64), (1024, 64)):               # the image still fits beside FOUR waves' 32-row tiles (round 4, second session)
        assert _lib.vq_kernel_name(K, D) == "vq_track_kernel_d64", (K, D)
        assert _lib.vq_sweeps(K, D) == 1
    for K, D in ((1025, 64), (3000, 64), (16384, 64), (64, 128), (8192, 128)):
        assert _lib.vq_kernel_name(K, D) == "vq_stream_sweep_kernel", (K, D)
        assert _lib.vq_sweeps(K, D) == 1
    assert _lib.vq_kernel_name(8192, 128, 0x1 | 0x4) == "vq_exact_kernel"      # VQVAE_VQ_EXACT_SWEEP
    assert _lib.vq_kernel_name(8192, 256) == "vq_exact_kernel"


@pytest.mark.parametrize("K,D,p,seed", [(640, 64, 11, 0), (1024, 64, 11, 1), (1024, 64, 11, 2), (2048, 128, 11, 3),
                                        (8192, 128, 11, 4), (96, 128, 11, 5), (3000, 64, 8, 6), (1056, 64, 11, 7), (2048, 64, 11, 8)])
def test_vq_stream_aligned_rounding_adversarial(K, D, p, seed):
    """tests/adversarial.py through the streamed-codebook kernel (vq_chunk.hip): several LDS chunks, several key
    epochs, D = 128, a K that is not a multiple of 32 -- every row bit for bit against the oracle."""
    from oracle import c_oracle
    from tests import adversarial as A
    zr, cb, _ = A.make_problem(p, K=K, D=D, n_rows=256, seed=seed)
    z = torch.from_numpy(np.ascontiguousarray(zr.reshape(4, 8, 8, D).transpose(0, 3, 1, 2)))
    ref = c_oracle.vq_forward(z.numpy(), cb, 0.25)
    for exact in (False, True):
        loss, zq, ppl, idx, hist = _run(z, torch.from_numpy(cb), 0.25, True, exact=exact)
        np.testing.assert_array_equal(idx, ref["idx"])
        assert np.array_equal(zq.view(np.uint32), ref["z_q"].view(np.uint32))
        np.testing.assert_allclose(loss, ref["loss"], rtol=1e-6)
        np.testing.assert_array_equal(hist, ref["hist"])


@pytest.mark.parametrize("K,D", [(1024, 64), (2048, 64), (1500, 128)])      # (K = 1024 at D = 64: the stream-tracker kernel's four-wave form)
def test_vq_stream_near_ties_duplicates_and_nonfinite_rows(K, D):
    """Everything the streamed kernel's task lists exist for: midpoints between codes of different key epochs (pair
    tasks), > 3 exact duplicates and near-duplicates (hard tasks), rows with NaN / Inf / beyond fp16's range (hard
    tasks with torch.argmin semantics), huge dynamic range in the codebook."""
    from oracle import c_oracle
    g = torch.Generator().manual_seed(K + D)
    cb = torch.randn(K, D, generator=g)
    cb[40:60] = cb[700]                                     # duplicates across epochs: first index must win
    cb[900:] *= torch.logspace(-2, 2, K - 900).unsqueeze(1)
    n = 1024
    ka = torch.randint(0, K, (n,), generator=g)
    kb = torch.randint(0, K, (n,), generator=g)
    zr = 0.5 * (cb[ka] + cb[kb]) + torch.randn(n, D, generator=g) * 1e-6
    zr[::5] = cb[ka[::5]] + torch.randn(len(ka[::5]), D, generator=g) * 1e-3
    zr[3::16] = cb[700]
    zr[7, 5] = float("nan")
    zr[64 + 9, 0] = float("inf")
    zr[200, 3] = -float("inf")
    zr[300] = 7.0e4                                         # finite, but not representable in fp16
    zr[301] = 1.0e-7                                        # fp16 subnormal / zero
    zr[302] = 0.0
    z = zr.view(n // 64, 8, 8, D).permute(0, 3, 1, 2).contiguous()
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    loss, zq, ppl, idx, hist = _run(z, cb, 0.25, True)
    np.testing.assert_array_equal(idx, ref["idx"])
    assert np.array_equal(np.isnan(zq), np.isnan(ref["z_q"]))
    m = ~np.isnan(zq)
    assert np.array_equal(zq[m].view(np.uint32), ref["z_q"][m].view(np.uint32))
    np.testing.assert_array_equal(hist, ref["hist"])


def test_vq_stream_two_slabs_ragged_vs_oracle():
    """More rows than one slab of the streamed kernel (2^18): 262 144 + 1 280 rows, K = 640."""
    g = torch.Generator().manual_seed(99)
    K, D, B, H, W = 640, 64, 4116, 8, 8                      # 263 424 rows
    cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
    z = torch.randn(B, D, H, W, generator=g) * 0.066
    ref_idx, ref_zq = _oracle_vq_chunked(z, cb, 0.25)
    loss, zq, ppl, idx, hist = _run(z, cb, 0.25, True)
    np.testing.assert_array_equal(idx, ref_idx)
    assert np.array_equal(zq.view(np.uint32), ref_zq.view(np.uint32))
    np.testing.assert_array_equal(hist, np.bincount(ref_idx.reshape(-1), minlength=K))
    a = _run(z, cb, 0.25, True)
    assert a[0].tobytes() == loss.tobytes() and a[2].tobytes() == ppl.tobytes()        # run-to-run bitwise


def test_vq_stream_more_than_one_group_of_slabs_equals_the_exhaustive_kernel():
    """The streamed kernels resolve the open / hard rows of sixteen slabs (2^22 rows) in one launch and start over for the
    next group: 17 slabs + a ragged tail, on the device only (1.1 GB of rows), indices / z_q / histogram / loss bits
    against the exhaustive fp32 kernel.  A few rows are made hard (many codes within the bound) in BOTH groups."""
    from vqvae_amd import functional as F
    g = torch.Generator(device=_dev()).manual_seed(5)
    K, D = 640, 64
    n = 17 * (1 << 18) + 333
    cb = ((torch.rand(K, D, generator=g, device=_dev()) * 2 - 1) / K)
    z = torch.randn(n // 3, 1, 3, D, generator=g, device=_dev()) * 0.066              # (B, H, W, D) row-major, 3 rows per "image"
    rows = z.view(-1, D)
    rows[12345] = cb[3:13].mean(0)                          # near-ties among many codes
    rows[(1 << 22) + 77] = cb[100:110].mean(0)
    rows[(1 << 22) + (1 << 18) + 5] = 0.0
    a = F.vq_forward(z, cb, 0.25, rowmajor=True)
    b = F.vq_forward(z, cb, 0.25, rowmajor=True, exact_sweep=True)
    torch.cuda.synchronize()
    assert torch.equal(a[3], b[3]), "indices"
    assert torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)), "z_q bits"
    assert torch.equal(a[4], b[4]), "histogram"
    assert a[0].item() == b[0].item() and a[2].item() == b[2].item()


def test_vq_nonfinite_codebook_forces_slow_path():
    """A codebook norm that is not < 1e38 routes EVERY row through the scalar torch.argmin path."""
    from oracle import c_oracle
    g = torch.Generator().manual_seed(3)
    cb = torch.randn(64, 64, generator=g)
    cb[5, 3] = float("inf")
    cb[9, 0] = float("nan")
    cb[11, :] = 2.0e19
    z = torch.randn(2, 64, 8, 8, generator=g)
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    loss, zq, ppl, idx, hist = _run(z, cb, 0.25)
    np.testing.assert_array_equal(idx, ref["idx"])
    assert np.array_equal(np.isnan(zq), np.isnan(ref["z_q"]))
    m = ~np.isnan(zq)
    assert np.array_equal(zq[m].view(np.uint32), ref["z_q"][m].view(np.uint32))


def test_vq_index_only_and_determinism():
    z, cb, beta = cases.vq_inputs("k512_d64_c1")
    a = _run(z, cb, beta, want_zq=False)
    b = _run(z, cb, beta)
    c = _run(z, cb, beta)
    assert a[1] is None
    np.testing.assert_array_equal(a[3], b[3])
    assert np.array_equal(b[1].view(np.uint32), c[1].view(np.uint32))
    assert b[0].tobytes() == c[0].tobytes() and b[2].tobytes() == c[2].tobytes()   # run-to-run bitwise


def test_vq_large_roundtrip_properties():
    """BASELINE config-3 size (262144 rows): size-independent properties.
    decode_indices(idx) must equal the codebook rows; z_q must equal fl(z + fl(e - z));
    hist sums to N; quantising the quantised output is idempotent on the indices."""
    from vqvae_amd import functional as F
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    K, D, B, H, W = 512, 64, 4096, 8, 8
    cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K).to(dev)
    z = (torch.randn(B, D, H, W, generator=g) * 0.066).to(dev)
    loss, zq, ppl, idx, hist = F.vq_forward(z, cb, 0.25)
    assert int(hist.sum()) == B * H * W
    e = F.vq_decode_indices(idx, cb, B, H, W)
    assert torch.equal(e.permute(0, 2, 3, 1).reshape(-1, D), cb[idx.view(-1)])
    assert torch.equal(zq, z + (e - z))
    # mean((e-z)^2)*(1+beta)
    m = ((e - z).double() ** 2).mean()
    np.testing.assert_allclose(loss.item(), float(m + 0.25 * m), rtol=1e-6)
    # exhaustive fp64 check of optimality: chosen distance within fp32 noise of the true minimum
    zf = z.permute(0, 2, 3, 1).reshape(-1, D).double()
    d = (zf ** 2).sum(1, keepdim=True) + (cb.double() ** 2).sum(1) - 2 * zf @ cb.double().t()
    chosen = d.gather(1, idx)
    assert float((chosen - d.min(1, keepdim=True).values).max()) < 1e-6
    _, _, _, idx2, _ = F.vq_forward(e, cb, 0.25)
    # codes are their own nearest neighbour unless an exact duplicate precedes them
    assert float((idx2 == idx).float().mean()) > 0.999


def test_onehot_and_decode_indices():
    from vqvae_amd import functional as F
    dev = _dev()
    idx = torch.randint(0, 100, (105, 1), device=dev)
    oh = F.vq_onehot(idx, 100)
    ref = torch.zeros(105, 100, device=dev).scatter_(1, idx, 1)
    assert torch.equal(oh, ref)


def test_decode_indices_rejects_out_of_range_indices():
    """an index outside [0, K) must never read past the codebook.  The front end raises like the
    reference's embedding lookup; the C entry point itself writes NaN for that element."""
    from vqvae_amd import functional as F, _lib
    dev = _dev()
    cb = torch.randn(100, 64, device=dev)
    idx = torch.randint(0, 100, (2 * 4 * 4, 1), device=dev)
    idx[5] = 100
    with pytest.raises(IndexError):
        F.vq_decode_indices(idx, cb, 2, 4, 4)
    idx[5] = -1
    with pytest.raises(IndexError):
        F.vq_decode_indices(idx, cb, 2, 4, 4)
    out = torch.zeros(2, 64, 4, 4, device=dev)
    idx[5] = 1 << 40
    _lib.check(_lib.load().vqvae_vq_decode_indices_f32(idx.data_ptr(), cb.data_ptr(), 2, 64, 4, 4, 100, out.data_ptr(),
                                                       torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = out.permute(0, 2, 3, 1).reshape(-1, 64)
    assert torch.isnan(o[5]).all() and torch.isfinite(o[torch.arange(32, device=dev) != 5]).all()
    idx[5] = 7
    assert torch.equal(F.vq_decode_indices(idx, cb, 2, 4, 4).permute(0, 2, 3, 1).reshape(-1, 64), cb[idx.view(-1)])


def test_codebook_cache_invalidation():
    """The prepared codebook image is keyed on (data_ptr, _version): a write through `.data` needs invalidate();
    in-place ops, load_state_dict and .to() are picked up automatically."""
    from vqvae_amd.modules import VectorQuantizer
    torch.manual_seed(1)
    vq = VectorQuantizer(64, 64, 0.25).to(_dev())
    z = torch.randn(2, 64, 4, 4, device=_dev())
    with torch.no_grad():
        a = vq(z)[4].clone()
        new = torch.randn(64, 64, device=_dev())
        vq.embedding.weight.data.copy_(new)              # does not bump _version
        vq.invalidate()
        b = vq(z)[4].clone()
        ref = torch.cdist(z.permute(0, 2, 3, 1).reshape(-1, 64), new).argmin(1, keepdim=True)
        assert torch.equal(b, ref) and not torch.equal(a, b)
        vq.embedding.weight.mul_(-1.0)                   # in-place op: picked up through _version
        c = vq(z)[4]
        assert torch.equal(c, torch.cdist(z.permute(0, 2, 3, 1).reshape(-1, 64), -new).argmin(1, keepdim=True))
        sd = {"embedding.weight": new.clone()}
        vq.load_state_dict(sd)
        vq.invalidate()
        assert torch.equal(vq(z)[4], b)


def test_module_interface_matches_reference_signature():
    from vqvae_amd.modules import VectorQuantizer
    torch.manual_seed(0)
    vq = VectorQuantizer(512, 64, 0.25).to(_dev())
    assert list(vq.state_dict().keys()) == ["embedding.weight"]
    z = torch.randn(4, 64, 8, 8, device=_dev()) * 0.066
    with torch.no_grad():
        loss, z_q, ppl, onehot, idx = vq(z)
        loss2, z_q2, *_ = vq(z)              # second call reuses the prepared codebook image
    assert loss.dim() == 0 and ppl.dim() == 0 and z_q.shape == z.shape
    assert onehot.shape == (256, 512) and idx.shape == (256, 1) and idx.dtype == torch.int64
    assert torch.equal(z_q, z_q2) and torch.equal(loss, loss2)
    with pytest.raises(Exception):
        vq(z.cpu())                          # no CPU fallback
    out = vq(z.clone().requires_grad_())     # under autograd: HIP forward + HIP backward (tests/test_training_gpu.py)
    assert out[0].requires_grad and out[1].requires_grad and torch.equal(out[1].detach(), z_q)


@pytest.mark.parametrize("K,D,N", [(1024, 64, 4096), (2048, 128, 2048), (8192, 128, 1024), (600, 64, 4096), (512, 32, 4096), (100, 256, 1024)])
@pytest.mark.parametrize("rowmajor", [True, False])
def test_vq_heterogeneous_rows_on_every_kernel_family(K, D, N, rowmajor):
    """Round 4's lesson (a row norm that was wrong only when a row's channels differ by decades) applied to the OTHER kernel
    families: the streamed-codebook kernels (K > ~600 or D = 128: BASELINE configs 4 / 5), the bf16 two-sweep filter and the
    exhaustive kernel's shapes -- rows whose channels carry independent 10^U(-3, 3) factors, a codebook of perturbed rows (every row
    has a near neighbour, many have several), against the C oracle bit for bit."""
    from oracle import c_oracle
    g = torch.Generator().manual_seed(K + D)
    f = 10.0 ** (torch.rand(D, generator=g) * 6 - 3)
    zr = torch.randn(N, D, generator=g) * f
    pick = torch.randint(0, N, (K,), generator=g)
    cb = zr[pick] * (1 + 1e-3 * torch.randn(K, D, generator=g))
    cb[::7] = zr[pick[::7]] + 1e-6 * f * torch.randn(len(pick[::7]), D, generator=g)      # some codes a hair away from a row
    z = zr.view(N // 64, 8, 8, D).permute(0, 3, 1, 2).contiguous()
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    loss, zq, ppl, idx, hist = _run(z, cb, 0.25, rowmajor)
    np.testing.assert_array_equal(idx, ref["idx"])
    assert np.array_equal(zq.view(np.uint32), ref["z_q"].view(np.uint32))
    np.testing.assert_allclose(loss, ref["loss"], rtol=1e-5)


def test_min_encodings_is_materialised_lazily():
    """`min_encodings` (the (N, K) one-hot of models/quantizer.py:55-57; 512 MiB at BASELINE config 3's size) is built
    on first use -- neither caller of the reference reads it (models/vqvae.py:34, visualization.ipynb:87).  The stand-in knows its
    shape / dtype / device without a launch and is the real tensor for everything else."""
    from vqvae_amd.modules import LazyOneHot, VectorQuantizer
    torch.manual_seed(3)
    vq = VectorQuantizer(96, 64, 0.25).to(_dev())
    z = torch.randn(4, 64, 8, 8, device=_dev()) * 0.01
    with torch.no_grad():
        loss, z_q, ppl, oh, idx = vq(z)
    assert isinstance(oh, LazyOneHot) and oh._t is None
    assert oh.shape == (256, 96) and oh.dtype == torch.float32 and oh.device == idx.device and len(oh) == 256 and oh.numel() == 256 * 96
    assert oh._t is None                                              # nothing above launched vqvae_vq_onehot_f32
    want = torch.zeros(256, 96, device=_dev()).scatter_(1, idx, 1)
    assert torch.equal(torch.matmul(oh, vq.embedding.weight.detach()), torch.matmul(want, vq.embedding.weight.detach()))   # a torch function
    assert oh._t is not None
    assert torch.equal(oh.cpu(), want.cpu()) and float(oh.sum()) == 256 and torch.equal(oh[3], want[3]) and torch.equal(oh * 2, want * 2)
    assert torch.equal(torch.mean(oh, dim=0), torch.mean(want, dim=0))
    vq.LAZY_MIN_ENCODINGS = False
    with torch.no_grad():
        oh2 = vq(z)[3]
    assert isinstance(oh2, torch.Tensor) and torch.equal(oh2, want)
