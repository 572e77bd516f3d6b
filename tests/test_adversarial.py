"""CPU tests of the aligned-rounding generators (tests/adversarial.py).

They prove that the inputs used by tests/test_vq_gpu.py::test_vq_aligned_rounding_adversarial really
are adversarial: a single-term 16-bit screen whose threshold assumes HALF the true unit roundoff
(round 1's u = 2^-9 for bf16) drops the reference's argmin (models/quantizer.py:49-54, restated by
oracle/vqvae_oracle.c) on every prototype row, and the sound threshold (2u + u^2) keeps it.
"""
import numpy as np
import pytest

from tests import adversarial as A


def _oracle_idx(z, cb):
    from oracle import c_oracle
    n, d = z.shape
    ref = c_oracle.vq_forward(np.ascontiguousarray(z.reshape(n, 1, 1, d).transpose(0, 3, 1, 2)), cb, 0.25)
    return ref["idx"].reshape(-1)


@pytest.mark.parametrize("p", [8, 11], ids=["bf16", "fp16"])
@pytest.mark.parametrize("seed", [0, 1, 4])
def test_pair_cases_defeat_a_half_sized_bound(p, seed):
    z, cb, k0 = A.make_problem(p, seed=seed)
    n_adv = z.shape[0] * 3 // 4
    idx = _oracle_idx(z, cb)
    assert np.isin(idx[:n_adv], [k0, k0 + 1]).all(), "fillers must never win a prototype row"
    u = 2.0 ** -p
    rows = np.arange(n_adv)
    tight = A.screen_candidates(z[:n_adv], cb, p, 1.0 * u)        # DELTA = 2 * (2 * u/2) |z| Emax: round 1
    sound = A.screen_candidates(z[:n_adv], cb, p, 2.0 * u + u * u)
    assert not tight[rows, idx[:n_adv]].any(), "the construction no longer defeats the half-sized bound"
    assert sound[rows, idx[:n_adv]].all(), "the sound bound must contain the reference's argmin"


@pytest.mark.parametrize("p", [8, 11], ids=["bf16", "fp16"])
@pytest.mark.parametrize("seed", range(6))
def test_sound_bound_contains_reference_argmin(p, seed):
    z, cb, _ = A.make_problem(p, seed=seed)
    idx = _oracle_idx(z, cb)
    u = 2.0 ** -p
    sound = A.screen_candidates(z, cb, p, 2.0 * u + u * u)
    assert sound[np.arange(z.shape[0]), idx].all()


