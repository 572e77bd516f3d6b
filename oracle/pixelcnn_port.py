"""ATen-op-for-op CPU restatement of the reference's GatedPixelCNN forward  --  TEST INFRASTRUCTURE ONLY.

Free functions over a state_dict issuing the same ATen ops in the same order as pixelcnn/models.py:29-135
(GatedActivation :21-27, GatedMaskedConv2d.forward :64-84 incl. make_causal :60-62, GatedPixelCNN.forward :118-127),
so on the host it runs on it is bitwise what the reference computes there.  Checked bitwise against the imported
reference in the build container (tests/test_oracle.py) and used to pin vqvae_amd/pixelcnn.py on the GPU.
Nothing under vqvae_amd/ imports it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def gate(x):
    a, b = x.chunk(2, dim=1)                                   # models.py:25-27
    return torch.tanh(a) * torch.sigmoid(b)


def forward(sd, x, label, n_layers):
    """sd: GatedPixelCNN.state_dict(); x (B,H,W) int64; label (B,) int64 -> logits (B, input_dim, H, W).
    Like the reference, the mask-A layer's weights are zeroed IN PLACE in `sd` (make_causal, :60-62)."""
    shp = x.size() + (-1,)
    t = F.embedding(x.view(-1), sd["embedding.weight"]).view(shp).permute(0, 3, 1, 2)      # :119-121
    x_v, x_h = t, t
    for i in range(n_layers):
        p = f"layers.{i}."
        k = 7 if i == 0 else 3
        if i == 0:                                             # mask 'A'
            sd[p + "vert_stack.weight"][:, :, -1].zero_()
            sd[p + "horiz_stack.weight"][:, :, :, -1].zero_()
        h = F.embedding(label, sd[p + "class_cond_embedding.weight"])                        # :68
        h_vert = F.conv2d(x_v, sd[p + "vert_stack.weight"], sd[p + "vert_stack.bias"], 1, (k // 2, k // 2))
        h_vert = h_vert[:, :, :x_v.size(-1), :]                                              # :70
        out_v = gate(h_vert + h[:, :, None, None])
        h_horiz = F.conv2d(x_h, sd[p + "horiz_stack.weight"], sd[p + "horiz_stack.bias"], 1, (0, k // 2))
        h_horiz = h_horiz[:, :, :, :x_h.size(-2)]                                            # :74
        v2h = F.conv2d(h_vert, sd[p + "vert_to_horiz.weight"], sd[p + "vert_to_horiz.bias"])
        out = gate(v2h + h_horiz + h[:, :, None, None])
        out_h = F.conv2d(out, sd[p + "horiz_resid.weight"], sd[p + "horiz_resid.bias"])
        if i > 0:                                              # residual = False only for the first layer (:106)
            out_h = out_h + x_h
        x_v, x_h = out_v, out_h
    t = F.relu(F.conv2d(x_h, sd["output_conv.0.weight"], sd["output_conv.0.bias"]))
    return F.conv2d(t, sd["output_conv.2.weight"], sd["output_conv.2.bias"])
