#!/usr/bin/env python3
"""Train a real checkpoint with the HIP training path  --  the loop of main.py:67-98 (Adam amsgrad lr 3e-4, model.train(),
recon / x_train_var + embedding loss, a random batch per update) on structured synthetic 32x32x3 images (tests/synthdata.py;
no dataset can be fetched here).  Every forward and backward runs on libvqvae_hip.so (vqvae_amd/autograd_conv.py,
training.VQStraightThrough, training.step_losses); the optimizer is torch's, as in the reference.

    python tools/train_checkpoint.py [--n_updates 5000] [--batch_size 32] [--out gpurun_out/trained]

Writes <out>/<tag>.pth in the reference's checkpoint layout (utils.py:109-113: {'model', 'results', 'hyperparameters'}),
<out>/<tag>_log.txt (the reference's log line every --log_interval updates + the range guard's per-layer spreads along the way;
committed as profiles/r06_train_*_log.txt) and <out>/<tag>_state.npz (the 23 state_dict tensors, fp32: what tests/golden/ keeps as
trained_main_defaults_state.npz [defaults] and trained_b128x20k_state.npz [--batch_size 128 --n_updates 20000]).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from tests import synthdata
from vqvae_amd import conv, training as T
from vqvae_amd.modules import VQVAE


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch_size", type=int, default=32)           # main.py:17-27's defaults
    p.add_argument("--n_updates", type=int, default=5000)
    p.add_argument("--n_hiddens", type=int, default=128)
    p.add_argument("--n_residual_hiddens", type=int, default=32)
    p.add_argument("--n_residual_layers", type=int, default=2)
    p.add_argument("--embedding_dim", type=int, default=64)
    p.add_argument("--n_embeddings", type=int, default=512)
    p.add_argument("--beta", type=float, default=.25)
    p.add_argument("--learning_rate", type=float, default=3e-4)
    p.add_argument("--log_interval", type=int, default=250)
    p.add_argument("--n_train", type=int, default=50000)            # CIFAR-10's training set size
    p.add_argument("--data_seed", type=int, default=2026)
    p.add_argument("--out", default="gpurun_out/trained")
    p.add_argument("--tag", default="vqvae_trained")
    args = p.parse_args()

    dev = torch.device("cuda:0")
    conv.set_conv_backend("hip")
    os.makedirs(args.out, exist_ok=True)
    log = open(os.path.join(args.out, f"{args.tag}_log.txt"), "w")

    def say(*a):
        s = " ".join(str(v) for v in a)
        print(s, flush=True)
        log.write(s + "\n")
        log.flush()

    t0 = time.time()
    x01 = synthdata.images01(args.n_train, args.data_seed)
    x_train_var = synthdata.train_var(x01)                          # utils.py:86
    data = ((x01 - 0.5) / 0.5).to(dev)                              # utils.py:15-16
    say(f"# data: {args.n_train} structured synthetic images (tests/synthdata.py seed {args.data_seed}), x_train_var={x_train_var:.6f}, "
        f"{time.time() - t0:.1f} s")
    say(f"# hyperparameters: {vars(args)}")

    torch.manual_seed(0)
    model = VQVAE(args.n_hiddens, args.n_residual_hiddens, args.n_residual_layers, args.n_embeddings, args.embedding_dim,
                  args.beta).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=args.learning_rate, amsgrad=True)       # main.py:59
    model.train()
    results = {"n_updates": 0, "recon_errors": [], "loss_vals": [], "perplexities": []}
    g = torch.Generator().manual_seed(1)
    stats_dev = []
    t0 = time.time()
    for i in range(args.n_updates):
        sel = torch.randint(0, args.n_train, (args.batch_size,), generator=g).to(dev)     # a shuffled loader's first batch
        x = data[sel].contiguous()
        opt.zero_grad()
        embedding_loss, x_hat, perplexity = model(x)
        stats = T.step_losses(embedding_loss, x_hat, perplexity, x, x_train_var)          # [recon_loss, loss, perplexity]
        stats[1].backward()                                                               # main.py:78
        opt.step()
        stats_dev.append(stats.detach())
        if i % args.log_interval == 0 or i == args.n_updates - 1:
            block = torch.stack(stats_dev).cpu().numpy()
            stats_dev = []
            results["recon_errors"] += block[:, 0].tolist()
            results["loss_vals"] += block[:, 1].tolist()
            results["perplexities"] += block[:, 2].tolist()
            results["n_updates"] = i
            model.eval()
            with torch.no_grad():
                rec, spreads = model.scheme_hint()
            model.train()
            say(f"Update # {i} Recon Error: {block[:, 0].mean():.5f} Loss {block[:, 1].mean():.5f} Perplexity: {block[:, 2].mean():.3f}"
                f"   | guard: flags={rec:#x} max spread {max(spreads):.2f} binades   [{time.time() - t0:.1f} s]")
    torch.cuda.synchronize()
    say(f"# {args.n_updates} updates in {time.time() - t0:.1f} s")

    model.eval()
    rec, spreads = model.scheme_hint()
    say(f"# final guard: flags={rec:#x}; per-layer input-channel spreads (binades): {[round(s, 2) for s in spreads]}")
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({"model": sd, "results": results, "hyperparameters": vars(args)}, os.path.join(args.out, f"{args.tag}.pth"))
    np.savez_compressed(os.path.join(args.out, f"{args.tag}_state.npz"), **{k: v.numpy() for k, v in sd.items()})

    # a first look at the checkpoint through the product path: validation images, all three product schemes
    from vqvae_amd import functional as F_hip
    xv = synthdata.normalised(4096, args.data_seed + 1).to(dev)
    with torch.no_grad():
        out = {}
        for name, fl in (("guard", None), ("fp16x2", 0), ("bf16x3", F_hip.FWD_CONV_BF16_SPLIT), ("fp32", F_hip.FWD_CONV_EXACT_FP32)):
            loss, x_hat, ppl, idx = model._forward_c(xv, want_idx=True, fwd_flags=fl)
            out[name] = (loss.item(), ppl.item(), idx.cpu(), x_hat.cpu())
            say(f"# eval[{name}]: embedding_loss={loss.item():.6g} perplexity={ppl.item():.4f} distinct codes={idx.unique().numel()} "
                f"recon mse/var={((x_hat - xv) ** 2).mean().item() / x_train_var:.5f}")
        for name in ("fp16x2", "bf16x3"):
            say(f"# index flips {name} vs fp32 scheme: {(out[name][2] != out['fp32'][2]).sum().item()} / {out['fp32'][2].numel()}; "
                f"max|x_hat diff| {(out[name][3] - out['fp32'][3]).abs().max().item():.3g}")
    log.close()


if __name__ == "__main__":
    main()
