"""CPU multi-process test (gloo, world_size 2) of the N>1 path: batch sharding + the optional
(K+1)-number merge of the batch-global scalars.  The per-shard quantizer outputs come from the
oracle (test infrastructure); the code under test is vqvae_amd/sharding.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import c_oracle
    from vqvae_amd import sharding
    z, cb, beta = cases.vq_inputs("k512_d64_c1")
    B, D, H, W = z.shape
    lo, hi = sharding.shard_bounds(B, world, rank)
    out = c_oracle.vq_forward(z[lo:hi].numpy(), cb.numpy(), beta)
    stats = sharding.vq_stats_from_outputs(torch.tensor(out["loss"]), torch.from_numpy(out["hist"]),
                                           (hi - lo) * H * W, D, beta)
    loss, ppl = sharding.merge_vq_stats(stats, B * H * W, D, beta)
    q.put((rank, lo, hi, float(loss), float(ppl), out["idx"].reshape(-1)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_reproduces_full_batch(golden_vq):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == 32       # contiguous cover
    idx = np.concatenate([r[5] for r in res])
    np.testing.assert_array_equal(idx, golden_vq["k512_d64_c1/idx"])           # shards == full batch rows
    for r in res:                                                              # merged scalars == reference's
        np.testing.assert_allclose(r[3], golden_vq["k512_d64_c1/loss"], rtol=1e-6)
        np.testing.assert_allclose(r[4], golden_vq["k512_d64_c1/perplexity"], rtol=1e-6)


def test_shard_bounds_cover():
    from vqvae_amd import sharding
    for n in (1, 7, 32, 4096, 8191):
        for w in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
