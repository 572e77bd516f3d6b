"""CPU emulation of the weight-gradient kernel's arithmetic (conv_wgrad_map8_h2_kernel, csrc/backward.hip; no GPU): two fp16 terms per
operand on ONE power-of-two scale per image and operand tile, the fp32 accumulator carried on the current image's scale and multiplied
by the exact power of two between images, images more than 2^60 below the largest one so far on a coarser scale (this emulation
found the first version's rule -- relative to the PREVIOUS image -- overflowing on a run of ever smaller images).  What is checked here is the
SCHEME (the statements of the kernel's header), on a 1x1 layer so that the sum is a plain matrix product:
  * well-scaled operands: error at the fp32 GEMM's own level;
  * images 10^12 apart in both directions, an all-zero image, one below the 2^-60 cut, the first image small: still there;
  * the accumulator never overflows and the rescaling is exact (powers of two only).
The GPU counterpart is tests/test_training_gpu.py::test_two_term_weight_gradient_across_image_scales."""
import numpy as np


def _scale_exp(m):
    if not (m > 0.0 and m < 3.0e38):
        return 0
    e = int(np.frexp(np.float32(m))[1])
    return max(-100, min(100, 15 - e))


def _split(x, k):
    xs = (x.astype(np.float64) * 2.0 ** k).astype(np.float32)
    h1 = xs.astype(np.float16)
    h2 = (xs - h1.astype(np.float32)).astype(np.float16)
    assert np.isfinite(h1.astype(np.float32)).all()
    return h1.astype(np.float64), h2.astype(np.float64)


def _emulate(a, b):
    """a (B, P, CA), b (B, P, CB) fp32 -> sum_b a_b^T b_b as the kernel forms it (fp32 accumulator, 16 pixels per MFMA step)."""
    B, P, CA = a.shape
    acc = np.zeros((CA, b.shape[2]), np.float32)
    e_acc, e_min, first = 0, 0, True
    for i in range(B):
        ka, kb = _scale_exp(np.abs(a[i]).max()), _scale_exp(np.abs(b[i]).max())
        e_min = ka + kb if first else min(e_min, ka + kb)
        if ka + kb > e_min + 60:                                       # (relative to the LARGEST image so far, not to the previous one)
            kb = e_min + 60 - ka
        a1, a2 = _split(a[i], ka)
        b1, b2 = _split(b[i], kb)
        e_img = ka + kb
        if first:
            e_acc, first = e_img, False
        elif e_img != e_acc:
            d = e_img - e_acc
            f = np.float32(0.0) if d < -120 else np.float32(2.0 ** d)
            before = acc.copy()
            acc = acc * f
            assert np.isfinite(acc).all()
            if f != 0 and abs(d) < 100:
                back = (acc.astype(np.float64) / float(f))
                ok = (back == before.astype(np.float64)) | (np.abs(acc) < 1e-37)        # exact unless it fell into the subnormals
                assert ok.all()
            e_acc = e_img
        for p0 in range(0, P, 16):                                    # one MFMA step = 16 pixels, three term products, fp32 accumulate
            sl = slice(p0, p0 + 16)
            for x, y in ((a2, b1), (a1, b2), (a1, b1)):
                acc = (acc.astype(np.float64) + x[sl].T @ y[sl]).astype(np.float32)
        assert np.isfinite(acc).all()
    return np.ldexp(acc.astype(np.float64), -e_acc)


def _case(fa, fb, seed=0, B=24, P=64, CA=32, CB=32):
    g = np.random.default_rng(seed)
    a = g.standard_normal((B, P, CA)).astype(np.float32) * np.asarray(fa, np.float32)[:, None, None]
    b = g.standard_normal((B, P, CB)).astype(np.float32) * np.asarray(fb, np.float32)[:, None, None]
    ref = np.einsum("bpi,bpj->ij", a.astype(np.float64), b.astype(np.float64))
    f32 = np.zeros((CA, CB), np.float32)
    for i in range(B):
        f32 = f32 + a[i].T @ b[i]
    got = _emulate(a, b)
    scale = np.abs(ref).max()
    return np.abs(got - ref).max() / scale, np.abs(f32.astype(np.float64) - ref).max() / scale


def test_well_scaled_operands_err_like_the_fp32_gemm():
    e, f = _case(np.ones(24), np.ones(24))
    assert e < 2e-6 and e < 8 * f + 1e-7, (e, f)


def test_images_decades_apart_zero_images_and_the_cut():
    g = np.random.default_rng(5)
    fa = 10.0 ** (g.random(24) * 12 - 6)
    fb = 10.0 ** (g.random(24) * 12 - 6)
    fa[3] = fb[5] = 0.0
    fa[7] = fb[7] = 1e-18                    # far below everything summed before it: coarser scale, negligible contribution
    fa[0] = fb[0] = 1e-6                     # the FIRST image small: the accumulator follows upwards
    e, f = _case(fa, fb, seed=1)
    assert e < 2e-6 and e < 8 * f + 1e-7, (e, f)
    # and the other order: the largest image first, then ever smaller ones
    e, f = _case(np.sort(fa)[::-1], np.sort(fb)[::-1], seed=2)
    assert e < 2e-6 and e < 8 * f + 1e-7, (e, f)
