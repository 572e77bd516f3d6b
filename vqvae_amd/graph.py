"""hipGraph replay of the forward path for launch-bound batch sizes.

At BASELINE config 1 (B=32) the ~20 kernels of `VQVAE.forward` run for a few tens of microseconds in total,
less than the host needs to launch them one by one.  `GraphedForward` captures one forward on a private
stream into a hipGraph (through `torch.cuda.CUDAGraph`, i.e. hipStreamBeginCapture / hipGraphLaunch on ROCm)
and replays it with a single launch; every kernel in the graph is still ours (libvqvae_hip.so), torch only
provides the capture, the static buffers and the stream.

    g = GraphedForward(model, example_x)         # warms up, captures
    embedding_loss, x_hat, perplexity = g(x)     # copies x into the static input, replays, returns the
                                                 # static outputs (valid until the next call; .clone() to keep)

Shapes are fixed at capture time; weights are read through their device pointers, so in-place weight updates
are seen by later replays, but the packed-weight / codebook images are only rebuilt by an eager call
(re-capture after changing parameters).  Forward-only, CUDA(HIP) fp32 tensors only, no fallback.
"""
from __future__ import annotations

import torch

from . import _lib


class GraphedForward:
    def __init__(self, model, example_x, warmup: int = 3):
        if not example_x.is_cuda:
            raise _lib.VqvaeHipError("GraphedForward needs a CUDA(HIP) example input: there is no CPU path")
        _lib.load()
        _lib.profile_enable(False)               # event records are not capturable work we want in the graph
        self.model = model
        self.static_x = example_x.detach().clone().contiguous()
        self._stream = torch.cuda.Stream(device=example_x.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad():
            # eager warm-up on the capture stream: packs weights, prepares the codebook image, sets kernel
            # attributes and fills the allocator's pools, none of which may happen inside the capture
            self._stream.wait_stream(torch.cuda.current_stream(example_x.device))
            with torch.cuda.stream(self._stream):
                for _ in range(max(1, warmup)):
                    out = model(self.static_x)
            self._stream.synchronize()
            with torch.cuda.graph(self._graph, stream=self._stream):
                out = model(self.static_x)
        self.static_out = out

    def __call__(self, x):
        if x.shape != self.static_x.shape or x.dtype != self.static_x.dtype:
            raise ValueError(f"graph captured for {tuple(self.static_x.shape)} {self.static_x.dtype}, "
                             f"got {tuple(x.shape)} {x.dtype}")
        self.static_x.copy_(x, non_blocking=True)
        self._graph.replay()
        return self.static_out

    def replay(self):
        """Replay on whatever is in `static_x` (no input copy)."""
        self._graph.replay()
        return self.static_out
