#!/usr/bin/env python3
"""Run the REAL reference GatedPixelCNN (imported from /root/reference, build container only) on seeded inputs and
commit its logits as tests/golden/pixelcnn_cases.npz, with the state_dict key list and a hash of its values: the
mirror module rebuilt from the same seed must reproduce both (torch's CPU generator is deterministic per version)."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
from pixelcnn.models import GatedPixelCNN          # noqa: E402

CASES = {"k512_dim64_l15": (512, 64, 15, 10, 4, 8, 8), "k64_dim32_l3": (64, 32, 3, 5, 3, 6, 6)}


def inputs(name):
    K, dim, nl, ncls, B, H, W = CASES[name]
    g = torch.Generator().manual_seed(77 + len(name))
    x = torch.randint(0, K, (B, H, W), generator=g)
    label = torch.randint(0, ncls, (B,), generator=g)
    return x, label


if __name__ == "__main__":
    out = {}
    for name, (K, dim, nl, ncls, B, H, W) in CASES.items():
        torch.manual_seed(0)
        m = GatedPixelCNN(K, dim, nl, ncls).eval()
        # the reference initialises biases to 0 and embeddings N(0,1); make biases non-trivial for the test
        with torch.no_grad():
            for n_, p in m.named_parameters():
                if n_.endswith("bias"):
                    p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n_))) * 0.05)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        x, label = inputs(name)
        with torch.no_grad():
            logits = m(x, label)
        out[f"{name}/logits"] = logits.numpy()
        out[f"{name}/sd_sha"] = np.frombuffer(hashlib.sha256(b"".join(
            sd0[k].numpy().tobytes() for k in sorted(sd0))).digest()[:8], dtype=np.uint8)
        out[f"{name}/sd_keys"] = np.array(list(sd0.keys()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pixelcnn_cases.npz"), **out)
    print("wrote", len(out), "arrays")
