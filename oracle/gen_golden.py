#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing and executing the UNMODIFIED reference
(/root/reference, read-only) in the build container.  TEST INFRASTRUCTURE ONLY.

The reference publishes no golden vectors for this path, so
the reference itself, run here on seeded inputs (tests/cases.py), is the anchor:
    python oracle/gen_golden.py [vq] [models] [trained]   # writes tests/golden/{vq_cases,model_cases,trained_cases}.npz
The reference tree cannot travel to the GPU box; these fixtures do.
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VQVAE_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np
import torch

import models.quantizer as ref_q                 # noqa: E402  (reference)
from models.quantizer import VectorQuantizer     # noqa: E402
from models.vqvae import VQVAE                   # noqa: E402

from tests import cases                          # noqa: E402

# module-global device (models/quantizer.py:7) must be CPU for a CPU model
ref_q.device = torch.device("cpu")
torch.set_num_threads(1)                         # bits are thread-count invariant


def gen_vq():
    out = {}
    for name, (K, D, B, H, W, beta, kind) in cases.VQ_CASES.items():
        z, cb, beta = cases.vq_inputs(name)
        vq = VectorQuantizer(K, D, beta)
        with torch.no_grad():
            vq.embedding.weight.copy_(cb)
            loss, z_q, ppl, onehot, idx = vq(z)
        assert onehot.shape == (B * H * W, K) and idx.shape == (B * H * W, 1)
        out[f"{name}/idx"] = idx.numpy().astype(np.int32).reshape(-1)
        out[f"{name}/loss"] = loss.numpy()
        out[f"{name}/perplexity"] = ppl.numpy()
        out[f"{name}/z_q"] = z_q.numpy() if z_q.numel() <= 70000 else np.zeros(0, np.float32)
        out[f"{name}/sha"] = np.array([cases.sha(z), cases.sha(cb), cases.sha(z_q),
                                       cases.sha(idx)])
        print(f"vq  {name:18s} N={B*H*W:6d} loss={loss.item():.9g} ppl={ppl.item():.9g} "
              f"distinct={idx.unique().numel()}")
    np.savez_compressed(os.path.join(ROOT, "tests/golden/vq_cases.npz"), **out)


def gen_models():
    out = {}
    for name, (h, rh, nl, K, D, beta, B, H, W) in cases.MODEL_CASES.items():
        torch.manual_seed(0)
        m = VQVAE(h, rh, nl, K, D, beta).eval()
        x = cases.model_inputs(name)
        with torch.no_grad():
            loss, x_hat, ppl = m(x)
            z_e = m.pre_quantization_conv(m.encoder(x))
            _, z_q, _, _, idx = m.vector_quantization(z_e)
            x_hat2 = m.decoder(z_q)
        assert torch.equal(x_hat, x_hat2)
        sd = m.state_dict()
        out[f"{name}/idx"] = idx.numpy().astype(np.int32).reshape(-1)
        out[f"{name}/loss"] = loss.numpy()
        out[f"{name}/perplexity"] = ppl.numpy()
        out[f"{name}/z_e"] = z_e.numpy()
        out[f"{name}/x_hat"] = x_hat.numpy()
        out[f"{name}/sha"] = np.array([cases.sha(x), cases.sha(z_e), cases.sha(z_q),
                                       cases.sha(x_hat), cases.sha(idx)] +
                                      [cases.sha(v) for v in sd.values()])
        out[f"{name}/keys"] = np.array(list(sd.keys()))
        print(f"mdl {name:18s} loss={loss.item():.18g} ppl={ppl.item():.18g} "
              f"idx.sum={int(idx.sum())} x_hat.sum={x_hat.double().sum().item():.12f}")
    np.savez_compressed(os.path.join(ROOT, "tests/golden/model_cases.npz"), **out)


def gen_trained():
    """Round 6: REALLY TRAINED checkpoints (tools/train_checkpoint.py: main.py:67-98's loop on the HIP training path, run on the
    GPU box; the 23 state_dict tensors are committed as tests/golden/<name>_state.npz) loaded into the UNMODIFIED reference model
    and run on seeded structured validation images (tests/synthdata.py).  -> tests/golden/trained_cases.npz"""
    from tests import synthdata
    out = {}
    for name, (h, rh, nl, K, D, beta, B, seed) in cases.TRAINED_CASES.items():
        sd = cases.trained_state(name)
        m = VQVAE(h, rh, nl, K, D, beta).eval()
        missing = m.load_state_dict(sd, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        x = synthdata.normalised(B, seed)
        with torch.no_grad():
            loss, x_hat, ppl = m(x)
            z_e = m.pre_quantization_conv(m.encoder(x))
            _, z_q, _, _, idx = m.vector_quantization(z_e)
            x_hat2 = m.decoder(z_q)
        assert torch.equal(x_hat, x_hat2)
        # (the images are committed too: bicubic / sin / exp of tests/synthdata.py round differently on another host's ISA)
        out[f"{name}/x"] = x.numpy()
        out[f"{name}/idx"] = idx.numpy().astype(np.int32).reshape(-1)
        out[f"{name}/loss"] = loss.numpy()
        out[f"{name}/perplexity"] = ppl.numpy()
        out[f"{name}/z_e"] = z_e.numpy()
        out[f"{name}/x_hat"] = x_hat.numpy()
        out[f"{name}/sha"] = np.array([cases.sha(x), cases.sha(z_e), cases.sha(z_q), cases.sha(x_hat), cases.sha(idx)])
        print(f"trn {name:22s} loss={loss.item():.9g} ppl={ppl.item():.9g} distinct={idx.unique().numel()} "
              f"max|z_e|={z_e.abs().max().item():.4g} max|x_hat|={x_hat.abs().max().item():.4g} recon mse={((x_hat - x) ** 2).mean().item():.5f}")
    np.savez_compressed(os.path.join(ROOT, "tests/golden/trained_cases.npz"), **out)


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests/golden"), exist_ok=True)
    which = sys.argv[1:] or ["vq", "models", "trained"]
    if "vq" in which:
        gen_vq()
    if "models" in which:
        gen_models()
    if "trained" in which:
        gen_trained()
