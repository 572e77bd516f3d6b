"""ATen-op-for-op CPU restatement of the reference forward  --  TEST INFRASTRUCTURE ONLY.

The reference's arithmetic lives in PyTorch's CPU kernels.  This
module restates `VQVAE.forward` as free functions over a state_dict, issuing the
same ATen ops in the same order as the reference modules, so that on the host it
runs on it is bitwise what the reference would compute there.  It is used
  * as the multi-core CPU baseline timed by bench.py (`cpu_baseline.kind="port"`;
    the Python reference itself cannot travel to the GPU box), and
  * as a second checker next to the C oracle (tests/test_oracle.py
    proves it bitwise-equal to the imported reference in the build container).
Nothing under vqvae_amd/ imports it.

Reference sites: models/vqvae.py:29-44, models/encoder.py:28-43,
models/residual.py:18-29,44-51, models/quantizer.py:45-76, models/decoder.py:27-39.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

E = "encoder.conv_stack."
Dk = "decoder.inverse_conv_stack."


def residual_stack(t, w1, w2, n_layers):
    """models/residual.py:47-51 incl. the aliasing (:44-45) and in-place ReLU (:19)."""
    for _ in range(n_layers):
        t = F.relu_(t)                                   # nn.ReLU(True) mutates the skip too
        h = F.conv2d(t, w1, None, 1, 1)                  # :20-21
        h = F.relu_(h)                                   # :22
        t = t + F.conv2d(h, w2, None, 1, 0)              # :23-24, :28
    return F.relu(t)                                     # :50


def encode(sd, x, n_res_layers):
    """models/encoder.py:28-43 then models/vqvae.py:33."""
    t = F.relu(F.conv2d(x, sd[E + "0.weight"], sd[E + "0.bias"], 2, 1))
    t = F.relu(F.conv2d(t, sd[E + "2.weight"], sd[E + "2.bias"], 2, 1))
    t = F.conv2d(t, sd[E + "4.weight"], sd[E + "4.bias"], 1, 1)
    t = residual_stack(t, sd[E + "5.stack.0.res_block.1.weight"],
                       sd[E + "5.stack.0.res_block.3.weight"], n_res_layers)
    return F.conv2d(t, sd["pre_quantization_conv.weight"], sd["pre_quantization_conv.bias"], 1, 0)


def quantize(z_e, codebook, beta):
    """models/quantizer.py:45-76 -> (loss, z_q, perplexity, min_encodings, idx)."""
    K, D = codebook.shape
    z = z_e.permute(0, 2, 3, 1).contiguous()
    zf = z.view(-1, D)
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1) \
        - 2 * torch.matmul(zf, codebook.t())
    idx = torch.argmin(d, dim=1).unsqueeze(1)
    onehot = torch.zeros(idx.shape[0], K)
    onehot.scatter_(1, idx, 1)
    z_q = torch.matmul(onehot, codebook).view(z.shape)
    loss = torch.mean((z_q - z) ** 2) + beta * torch.mean((z_q - z) ** 2)
    z_q = z + (z_q - z)
    e_mean = torch.mean(onehot, dim=0)
    perplexity = torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10)))
    return loss, z_q.permute(0, 3, 1, 2).contiguous(), perplexity, onehot, idx


def quantize_train(z_e, codebook, beta):
    """models/quantizer.py:45-76 with its .detach() calls (:63-64, :67), for autograd: the gradients
    torch derives from this are the oracle of vqvae_vq_backward_f32.  Same forward values as quantize()."""
    K, D = codebook.shape
    z = z_e.permute(0, 2, 3, 1).contiguous()
    zf = z.view(-1, D)
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1) \
        - 2 * torch.matmul(zf, codebook.t())
    idx = torch.argmin(d, dim=1).unsqueeze(1)
    onehot = torch.zeros(idx.shape[0], K)
    onehot.scatter_(1, idx, 1)
    z_q = torch.matmul(onehot, codebook).view(z.shape)
    loss = torch.mean((z_q.detach() - z) ** 2) + beta * torch.mean((z_q - z.detach()) ** 2)
    z_q = z + (z_q - z).detach()
    e_mean = torch.mean(onehot, dim=0)
    perplexity = torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10)))
    return loss, z_q.permute(0, 3, 1, 2).contiguous(), perplexity, onehot, idx


def decode(sd, z_q, n_res_layers):
    """models/decoder.py:27-39."""
    t = F.conv_transpose2d(z_q, sd[Dk + "0.weight"], sd[Dk + "0.bias"], 1, 1)
    t = residual_stack(t, sd[Dk + "1.stack.0.res_block.1.weight"],
                       sd[Dk + "1.stack.0.res_block.3.weight"], n_res_layers)
    t = F.relu(F.conv_transpose2d(t, sd[Dk + "2.weight"], sd[Dk + "2.bias"], 2, 1))
    return F.conv_transpose2d(t, sd[Dk + "4.weight"], sd[Dk + "4.bias"], 2, 1)


@torch.no_grad()
def forward(sd, x, beta, n_res_layers, full=False):
    """models/vqvae.py:29-44 -> (embedding_loss, x_hat, perplexity) [+ z_e, z_q, idx]."""
    z_e = encode(sd, x, n_res_layers)
    loss, z_q, ppl, _, idx = quantize(z_e, sd["vector_quantization.embedding.weight"], beta)
    x_hat = decode(sd, z_q, n_res_layers)
    if full:
        return loss, x_hat, ppl, z_e, z_q, idx
    return loss, x_hat, ppl


def init_state_dict(h_dim=128, res_h_dim=32, n_embeddings=512, embedding_dim=64, in_ch=3,
                    seed=0, n_res_layers=2):
    """Random weights with the reference's default initialisers and in the
    reference's construction order (models/vqvae.py:15-22), so that
    `torch.manual_seed(seed)` followed by this call consumes the CPU generator
    exactly like `VQVAE(...)` does.  Used for synthetic benchmarks/tests only."""
    import torch.nn as nn
    torch.manual_seed(seed)
    mods = {
        E + "0": nn.Conv2d(in_ch, h_dim // 2, 4, 2, 1),
        E + "2": nn.Conv2d(h_dim // 2, h_dim, 4, 2, 1),
        E + "4": nn.Conv2d(h_dim, h_dim, 3, 1, 1),
        E + "5.stack.0.res_block.1": nn.Conv2d(h_dim, res_h_dim, 3, 1, 1, bias=False),
        E + "5.stack.0.res_block.3": nn.Conv2d(res_h_dim, h_dim, 1, 1, bias=False),
        "pre_quantization_conv": nn.Conv2d(h_dim, embedding_dim, 1, 1),
    }
    emb = nn.Embedding(n_embeddings, embedding_dim)
    emb.weight.data.uniform_(-1.0 / n_embeddings, 1.0 / n_embeddings)   # quantizer.py:27
    mods2 = {
        Dk + "0": nn.ConvTranspose2d(embedding_dim, h_dim, 3, 1, 1),
        Dk + "1.stack.0.res_block.1": nn.Conv2d(h_dim, res_h_dim, 3, 1, 1, bias=False),
        Dk + "1.stack.0.res_block.3": nn.Conv2d(res_h_dim, h_dim, 1, 1, bias=False),
        Dk + "2": nn.ConvTranspose2d(h_dim, h_dim // 2, 4, 2, 1),
        Dk + "4": nn.ConvTranspose2d(h_dim // 2, in_ch, 4, 2, 1),
    }
    sd = {}
    for name, m in list(mods.items()):
        for p, v in m.named_parameters():
            sd[f"{name}.{p}"] = v.detach()
    sd["vector_quantization.embedding.weight"] = emb.weight.detach()
    for name, m in mods2.items():
        for p, v in m.named_parameters():
            sd[f"{name}.{p}"] = v.detach()
    # the aliased residual layers 1..n-1 (models/residual.py:44-45)
    for pre in (E + "5.stack.", Dk + "1.stack."):
        for l in range(1, n_res_layers):
            for blk in ("res_block.1.weight", "res_block.3.weight"):
                sd[f"{pre}{l}.{blk}"] = sd[pre + "0." + blk]
    return sd
