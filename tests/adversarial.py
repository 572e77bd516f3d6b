"""Aligned-rounding inputs for the quantizer's low-precision screen (tests only).

The fused VectorQuantizer kernels screen `z . e_k - ||e_k||^2/2` on the 16-bit matrix cores and then
refine exactly; the screen is only sound if its threshold covers the WORST case of the operand
roundings, i.e. every one of the D channel roundings pushing the same way.  Random perturbations never
align 64 rounding errors, so these generators build the alignment on purpose (VERDICT.md round 1,
"What's weak" #1; models/quantizer.py:49-54 is the semantics that must survive):

  * every channel of z and of a code pair sits just above / just below a midpoint of the screen's
    number format (significand bits p: 8 = bf16, 11 = fp16), with signs chosen so that all roundings
    move the screened dot product the same way;
  * one free channel sets the TRUE fp32 margin between the two codes to a small value of the opposite
    sign, so the screen's favourite is not the reference's argmin.

`screen_candidates` is a numpy model of a single-term screen with threshold DELTA = 2 * c * u * |z| * Emax,
used by the CPU tests to prove that the cases do defeat a bound with c*u too small (e.g. u = 2^-9 for
bf16) and are contained by the sound one.
"""
from __future__ import annotations

import numpy as np


def _round_to(x, p):
    """round-to-nearest-even to p significand bits (no exponent-range effects), float64 in/out."""
    x = np.asarray(x, dtype=np.float64)
    m, e = np.frexp(x)                       # x = m * 2^e, 0.5 <= |m| < 1
    q = np.ldexp(np.rint(np.ldexp(m, p)), -p)
    return np.ldexp(q, e)


def fma_chain(z, w):
    """c-ordered fp32 fmaf chain (torch.matmul's order on CPU for D <= 256)."""
    acc = np.float32(0.0)
    for a, b in zip(np.asarray(z, np.float32), np.asarray(w, np.float32)):
        acc = np.float32(np.float64(a) * np.float64(b) + np.float64(acc))
    return acc


def pair_case(p: int, D: int = 64, margin: float = 1.0e-3, eta: float = 2.0 ** -14, flip: bool = False):
    """-> (z (D,), w (D,)) with codes e0 = -w, e1 = +w (equal norms).

    All D-1 channel roundings raise the screened z.w by ~2 * 2^-p each, while the true fp32 chain
    z.w is about -margin: the reference's argmin is code 0 (-w), the screen's favourite is code 1 (+w).
    `flip` mirrors the construction (reference picks code 1)."""
    half = D // 2
    a = 1.0 + 2.0 ** -p + eta                # rounds UP   to 1 + 2^-(p-1)
    b = 1.0 + 2.0 ** -p - eta                # rounds DOWN to 1
    z = np.empty(D, np.float64)
    w = np.empty(D, np.float64)
    z[:half - 1] = a;  w[:half - 1] = a      # true a^2,  screened (1+2^-(p-1))^2  -> larger
    z[half - 1:D - 1] = b;  w[half - 1:D - 1] = -b   # true -b^2, screened -1          -> larger
    z[D - 1] = 1.0
    z32, w32 = z.astype(np.float32), w.astype(np.float32)
    # free channel: make the fp32 chain land on -margin
    part = float(np.dot(z32[:-1].astype(np.float64), w32[:-1].astype(np.float64)))
    w32[D - 1] = np.float32(-margin - part)
    for _ in range(8):                       # the chain's own roundings move the result a little: correct
        got = float(fma_chain(z32, w32))
        w32[D - 1] = np.float32(float(w32[D - 1]) + (-margin - got))
    if flip:
        w32 = -w32
    return z32, w32


def split_support_case(p: int, D: int = 64, margin: float = 1.0e-3, eta: float = 2.0 ** -14):
    """z over two disjoint channel sets; on the first both z and code A round DOWN
    (A's score is under-estimated), on the second both z and code B round UP (B's is over-estimated).
    A free channel (last of A's set) makes A the true winner by ~margin.  -> (z, eA, eB)."""
    h = D // 2
    a = 1.0 + 2.0 ** -p + eta
    b = 1.0 + 2.0 ** -p - eta
    z = np.zeros(D); eA = np.zeros(D); eB = np.zeros(D)
    z[:h] = b;  eA[:h] = b                   # A lives on channels [0, h)
    z[h:] = a;  eB[h:] = a                   # B lives on channels [h, D)
    z, eA, eB = (v.astype(np.float32) for v in (z, eA, eB))

    def score(e):
        return float(fma_chain(z, e)) - 0.5 * float(np.sum(e.astype(np.float64) ** 2))
    for _ in range(12):                      # tune A's last channel until score(A) - score(B) ~ margin
        gap = score(eA) - score(eB)
        # d score / d eA[h-1] = z[h-1] - eA[h-1]; keep it away from 0 by stepping z's partner too
        eA[h - 1] = np.float32(float(eA[h - 1]) + (margin - gap) / max(1e-3, abs(float(z[h - 1]) - float(eA[h - 1])) + 0.5))
        z[h - 1] = np.float32(float(z[h - 1]) + 0.25 * (margin - gap))
    return z, eA, eB


def make_problem(p: int, K: int = 512, D: int = 64, n_rows: int = 256, seed: int = 0):
    """A whole quantizer problem built around ONE aligned-rounding code pair.

    -> (z_rows (n_rows, D) fp32, codebook (K, D) fp32, pair position k0).  Codes k0, k0+1 are the
    +-w pair (seed % 3 != 2) or the A/B pair (seed % 3 == 2), randomly sign-flipped per channel,
    channel-permuted and scaled by a power of two (all of which preserve the alignment).  The other
    K-2 codes are fillers that can never win (1.4x the pair norm, orthogonal to the prototype row) but
    keep max|e_k| close to the pair norm, so a threshold that is too tight by 2x still excludes the
    reference's argmin.  Rows: the prototype times powers of two (the +-w pair has equal norms, so
    its order does not depend on |z|), then ordinary random rows to exercise the common path."""
    rng = np.random.default_rng(seed)
    cb = np.zeros((K, D), np.float32)
    margin = [1e-3, 1e-2, 0.05, 0.2, 3e-4][seed % 5]
    sc_e = np.float32(2.0 ** int(rng.integers(-9, 3)))
    perm = rng.permutation(D)
    sgn = rng.choice([-1.0, 1.0], size=D).astype(np.float32)
    split = seed % 3 == 2
    if split:
        z, c0, c1 = split_support_case(p, D, margin=margin)
    else:
        z, w = pair_case(p, D, margin=margin, flip=bool(seed & 1))
        c0, c1 = -w, w
    proto = (z * sgn)[perm] * sc_e
    k0 = int(rng.integers(0, K - 1))
    cb[k0] = (c0 * sgn)[perm] * sc_e
    cb[k0 + 1] = (c1 * sgn)[perm] * sc_e
    pn = proto.astype(np.float64) / np.linalg.norm(proto.astype(np.float64))
    pair_norm = float(np.linalg.norm(cb[k0:k0 + 2].astype(np.float64), axis=1).max())
    for k in range(K):
        if k in (k0, k0 + 1):
            continue
        f = rng.standard_normal(D)
        f -= pn * (pn @ f)
        cb[k] = (f / np.linalg.norm(f) * 1.4 * pair_norm).astype(np.float32)
    zs = np.empty((n_rows, D), np.float32)
    n_adv = n_rows * 3 // 4
    for i in range(n_adv):
        zs[i] = proto * np.float32(1.0 if split else 2.0 ** ((i % 9) - 4))
    zs[n_adv:] = rng.standard_normal((n_rows - n_adv, D)).astype(np.float32) * sc_e * 8
    return zs, cb, k0


def screen_candidates(z, cb, p: int, c_u: float):
    """numpy model of a single-term screen: operands rounded to p significand bits, score
    s_k = zr . er_k - ||e_k||^2/2 in float64, candidates = {k : s_k >= max - 2 * c_u * |z| * Emax}.
    -> boolean (N, K)."""
    z = np.asarray(z, np.float64); cb = np.asarray(cb, np.float64)
    zr, er = _round_to(z, p), _round_to(cb, p)
    s = zr @ er.T - 0.5 * (cb ** 2).sum(1)[None]
    delta = 2.0 * c_u * np.linalg.norm(z, axis=1) * np.linalg.norm(cb, axis=1).max()
    return s >= (s.max(1) - delta)[:, None]
