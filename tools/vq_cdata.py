#!/usr/bin/env python3
"""z_e rows of the reference-initialised model (BASELINE config 3 dims) + the reference's indices, as one binary file for the
torch-free A/B harness tools/ubench/vq_ab.cpp:   int64 N, K, D | z (N, D) f32 | codebook (K, D) f32 | idx (N) int32.
Computed on the CPU by oracle/torch_port.py (test infrastructure; this tool is a measurement aid, not product code).
    python tools/vq_cdata.py [n_images=1024] [out=tools/data/vq_c3.bin] [K=512]      (tools/data/ is not tracked; it travels with gpurun)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_port as tp

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "vq_c3.bin")
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
sd = tp.init_state_dict(128, 32, K, 64, seed=0)
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    x = torch.randn(n_images, 3, 32, 32, generator=g)
    z_e = tp.encode(sd, x, 2)
    cb = sd["vector_quantization.embedding.weight"]
    _, _, _, _, idx = tp.quantize(z_e, cb, 0.25)
rows = z_e.permute(0, 2, 3, 1).contiguous().view(-1, 64).numpy()
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "wb") as f:
    np.array([rows.shape[0], cb.shape[0], cb.shape[1]], dtype=np.int64).tofile(f)
    rows.astype(np.float32).tofile(f)
    cb.numpy().astype(np.float32).tofile(f)
    idx.view(-1).numpy().astype(np.int32).tofile(f)
print(out, rows.shape, "distinct codes:", int(idx.unique().numel()))
