#!/usr/bin/env python3
"""Row classes of the fused quantizer's screen (closed / open / hard / non-finite) on the z_e of real models -- the default init on
N(0,1) images (SURVEY.md 7.2-H1 regime A, what the headline benchmark feeds) and the two trained checkpoints on structured images
(regime C).  Needs the VQ_DEBUG_VERDICT build, whose kernel writes every row's classification beside its index:

    python tools/build_variant.py dbg -DVQ_DEBUG_VERDICT
    VQVAE_HIP_LIB_OVERRIDE=vqvae_amd/build/variants/libvqvae_dbg.so python tools/r06_row_classes.py [out.json]

closed = exactly one code at or above the rigorous threshold: the screen alone decides; open = the top-2 streams x cells become exact
fp32 tasks; hard = the tile is screened again; bad = non-finite row, scalar torch.argmin path."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import cases, synthdata
from vqvae_amd import conv, conv_hip, functional as F
from vqvae_amd.modules import VQVAE

assert os.environ.get("VQVAE_HIP_LIB_OVERRIDE"), "run with the VQ_DEBUG_VERDICT variant (see the docstring)"
dev = torch.device("cuda:0")
conv.set_conv_backend("hip")
B = 4096
out = {}


def classes(model, x, beta=0.25):
    with torch.no_grad():
        z_e = conv_hip.encoder_forward(model.encoder, x, model.pre_quantization_conv)
        cb = model.vector_quantization.embedding.weight.detach()
        idx = F.vq_forward(z_e, cb, beta, rowmajor=True)[3]
    v = idx.view(-1).cpu().numpy()
    n = v.size
    fl = (v >> 20) & 0xF
    return {"rows": int(n), "open": float(((fl & 1) != 0).mean()), "hard": float(((fl & 2) != 0).mean()),
            "bad": float(((fl & 4) != 0).mean()), "closed": float((fl & 7 == 0).mean()),
            "codes_in_use": int(len(set((v & 0xFFFFF).tolist()))),
            "max_abs_z_e": float(z_e.abs().max()), "source": "tools/r06_row_classes.py, VQ_DEBUG_VERDICT build"}


torch.manual_seed(0)
m = VQVAE(128, 32, 2, 512, 64, 0.25).eval().to(dev)
out["default_init"] = classes(m, torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(1000)).to(dev))
for name, (h, rh, nl, K, D, beta, _, seed) in cases.TRAINED_CASES.items():
    m = VQVAE(h, rh, nl, K, D, beta).eval()
    m.load_state_dict(cases.trained_state(name))
    out[name] = classes(m.to(dev), synthdata.normalised(B, seed + 7).to(dev), beta)
for k, v in out.items():
    print(k, v)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
