"""Host-side mirror of the reference's GatedPixelCNN prior (pixelcnn/models.py) on libvqvae_hip.so.

Same class names, constructor signatures, sub-module / parameter names (hence `state_dict` keys) and default
initialisation as the reference, so a PixelCNN checkpoint written by `pixelcnn/gated_pixelcnn.py` loads unchanged:

    GatedPixelCNN(input_dim=256, dim=64, n_layers=15, n_classes=10)
        .forward(x (B,H,W) int64, label (B,) int64) -> logits (B, input_dim, H, W)          (models.py:118-127)
        .generate(label, shape=(8, 8), batch_size=64) -> (B, *shape) int64                   (models.py:129-142)

Forward-only.  Activations are row-major (B,H,W,C); a masked convolution is a stride-1 convolution over its causal tap list
(round 4: vqvae_conv_taps_forward_f32 -- the conv path's kernels with an explicit tap list, no im2col pass; lists of more than
16 taps, i.e. the first layer's 4 x 7 vertical stack, run as a chain of launches over slices of the list); embeddings, the gated
activation (+ class-conditional term) and the residual add are small HIP kernels (csrc/pixelcnn.hip).  The
categorical sampling of `generate` uses torch's softmax + multinomial on the device, as the reference does.
No CPU path, no fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _cache, _lib, conv_hip
from ._lib import VqvaeHipError
from .conv_hip import CONV_1x1, RELU_OUT, _sp


def weights_init(m):
    """models.py:9-16 (the reference prints when a module has no .weight; the RNG stream is what matters)."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        try:
            nn.init.xavier_uniform_(m.weight.data)
            m.bias.data.fill_(0)
        except AttributeError:
            pass


class GatedActivation(nn.Module):
    def forward(self, x):                                       # parameter-free; fused into the HIP kernels
        raise VqvaeHipError("GatedActivation is fused into GatedMaskedConv2d's HIP forward")


class _Holder:
    pass


def _gather_rows(idx, table):
    idx = idx.contiguous().view(-1)
    n, Cc = idx.numel(), table.shape[1]
    out = torch.empty((n, Cc), dtype=torch.float32, device=table.device)
    _lib.check(_lib.load().vqvae_gather_rows_f32(idx.data_ptr(), table.detach().contiguous().data_ptr(), n, Cc,
                                                 table.shape[0], out.data_ptr(), _sp(out)))
    return out


def _im2col(x_rows, taps):
    B, H, W, Cc = x_rows.shape
    n = len(taps)
    dy = (C.c_int8 * n)(*[t[0] for t in taps])
    dx = (C.c_int8 * n)(*[t[1] for t in taps])
    out = torch.empty((B, H, W, n * Cc), dtype=torch.float32, device=x_rows.device)
    _lib.check(_lib.load().vqvae_im2col_rows_f32(x_rows.data_ptr(), B, H, W, Cc, n, C.cast(dy, C.c_void_p),
                                                 C.cast(dx, C.c_void_p), out.data_ptr(), _sp(out)))
    return out


def _gate(t1, t2, cond, dim):
    B, H, W, _ = t1.shape
    out = torch.empty((B, H, W, dim), dtype=torch.float32, device=t1.device)
    _lib.check(_lib.load().vqvae_gated_activation_f32(t1.data_ptr(), t2.data_ptr() if t2 is not None else None,
                                                      cond.data_ptr() if cond is not None else None, B, H * W, dim,
                                                      out.data_ptr(), _sp(out)))
    return out


def _add(a, b):
    out = torch.empty_like(a)
    _lib.check(_lib.load().vqvae_add_f32(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), _sp(a)))
    return out


class GatedMaskedConv2d(nn.Module):
    """Mirrors pixelcnn/models.py:29-84."""

    def __init__(self, mask_type, dim, kernel, residual=True, n_classes=10):
        super().__init__()
        assert kernel % 2 == 1, "Kernel size must be odd"
        self.mask_type = mask_type
        self.residual = residual
        self.dim, self.kernel = dim, kernel
        self.class_cond_embedding = nn.Embedding(n_classes, 2 * dim)
        self.vert_stack = nn.Conv2d(dim, dim * 2, (kernel // 2 + 1, kernel), 1, (kernel // 2, kernel // 2))
        self.vert_to_horiz = nn.Conv2d(2 * dim, 2 * dim, 1)
        self.horiz_stack = nn.Conv2d(dim, dim * 2, (1, kernel // 2 + 1), 1, (0, kernel // 2))
        self.horiz_resid = nn.Conv2d(dim, dim, 1)
        self.gate = GatedActivation()
        k = kernel
        # tap lists in the (ky, kx) order of the weight tensors; the conv's crop (:70, :74) keeps rows / columns
        # y + ky - k//2 and x + kx - k//2
        self._vtaps = [(ky - k // 2, kx - k // 2) for ky in range(k // 2 + 1) for kx in range(k)]
        self._htaps = [(0, kx - k // 2) for kx in range(k // 2 + 1)]

    def make_causal(self):                                      # models.py:60-62 (in place, like the reference)
        # through the parameter under no_grad, NOT `.data`: a `.data` write leaves `_version` alone, and the packed-weight cache
        # (conv_hip._packed, keyed on (data_ptr, _version)) would keep an image packed from un-masked or re-initialised weights
        #  -- e.g. after the reference's own `weights_init`, which writes through `.data`
        # The write bumps the version on EVERY call, so the one mask-'A' layer re-packs per forward (two small pack launches, also
        # inside a stream capture) -- the price of never serving a stale image for the layer whose weights the forward itself edits.
        with torch.no_grad():
            self.vert_stack.weight[:, :, -1].zero_()
            self.horiz_stack.weight[:, :, :, -1].zero_()

    def _masked(self, x_rows, conv, taps, tag):
        """One masked conv (bias included): the conv kernels over the tap list."""
        # (mask 'A': make_causal's in-place zeroing bumps the version and re-packs, conv_hip._packed; its 4 x 7 vertical stack is two
        # slices of 14 taps, the second launch adding to the first)
        return conv_hip.conv_taps(x_rows, conv, conv.weight, conv.bias, taps)

    def forward_rows(self, x_v, x_h, label):
        """x_v, x_h row-major (B,H,W,dim); label (B,) int64 -> (out_v, out_h) row-major."""
        if self.mask_type == 'A':
            self.make_causal()
        dim = self.dim
        cond = _gather_rows(label, self.class_cond_embedding.weight)                       # :68
        h_vert = self._masked(x_v, self.vert_stack, self._vtaps, "vert")                   # :69-70
        out_v = _gate(h_vert, None, cond, dim)                                              # :71
        h_horiz = self._masked(x_h, self.horiz_stack, self._htaps, "horiz")                # :73-74
        # :75, :77  v2h + h_horiz in the 1x1 conv's epilogue (same sum, same order: (v2h + h_horiz) + cond in the gate)
        v2h = conv_hip.conv(CONV_1x1, h_vert, self.vert_to_horiz, self.vert_to_horiz.weight, self.vert_to_horiz.bias,
                            2 * dim, 2 * dim, 0, addend=h_horiz)
        out = _gate(v2h, None, cond, dim)
        # :78-79  the horizontal residual in the 1x1 conv's epilogue
        out_h = conv_hip.conv(CONV_1x1, out, self.horiz_resid, self.horiz_resid.weight, self.horiz_resid.bias, dim,
                              dim, 0, addend=x_h.contiguous() if self.residual else None)
        return out_v, out_h

    def forward(self, x_v, x_h, h):
        """NCHW boundary of the reference (models.py:64-84)."""
        ov, oh = self.forward_rows(conv_hip.nchw_to_rows(x_v), conv_hip.nchw_to_rows(x_h), h)
        return conv_hip.rows_to_nchw(ov), conv_hip.rows_to_nchw(oh)


class GatedPixelCNN(nn.Module):
    """Mirrors pixelcnn/models.py:87-142."""

    def __init__(self, input_dim=256, dim=64, n_layers=15, n_classes=10):
        super().__init__()
        self.dim = dim
        self.embedding = nn.Embedding(input_dim, dim)
        self.layers = nn.ModuleList()
        for i in range(n_layers):
            mask_type = 'A' if i == 0 else 'B'
            kernel = 7 if i == 0 else 3
            residual = False if i == 0 else True
            self.layers.append(GatedMaskedConv2d(mask_type, dim, kernel, residual, n_classes))
        self.output_conv = nn.Sequential(nn.Conv2d(dim, 512, 1), nn.ReLU(True), nn.Conv2d(512, input_dim, 1))
        self.apply(weights_init)

    @torch.no_grad()
    def forward(self, x, label):
        if not x.is_cuda or x.dtype != torch.int64:
            raise VqvaeHipError("GatedPixelCNN.forward needs CUDA(HIP) int64 indices: there is no CPU path")
        if self.dim % 4:
            raise VqvaeHipError("dim must be a multiple of 4 for the HIP kernels")
        B, H, W = x.shape
        t = _gather_rows(x, self.embedding.weight).view(B, H, W, self.dim)                  # :119-121
        x_v, x_h = t, t
        for layer in self.layers:
            x_v, x_h = layer.forward_rows(x_v, x_h, label)
        c0, c2 = self.output_conv[0], self.output_conv[2]
        t = conv_hip.conv(CONV_1x1, x_h, c0, c0.weight, c0.bias, self.dim, c0.weight.shape[0], RELU_OUT)
        t = conv_hip.conv(CONV_1x1, t, c2, c2.weight, c2.bias, c0.weight.shape[0], c2.weight.shape[0], 0)
        return conv_hip.rows_to_nchw(t)                                                     # (B, input_dim, H, W)

    @torch.no_grad()
    def generate(self, label, shape=(8, 8), batch_size=64, use_graph=False):
        """models.py:129-142: one full forward per position, categorical sample from the softmax.

        use_graph=True captures the forward once into a hipGraph and replays it for each of the H*W positions:
        a forward is ~110 small launches, so at sampling batch sizes the eager loop is launch-bound."""
        param = next(self.parameters())
        x = torch.zeros((batch_size, *shape), dtype=torch.int64, device=param.device)
        label = label.contiguous()
        if use_graph:
            _lib.profile_enable(False)
            stream = torch.cuda.Stream(device=param.device)
            stream.wait_stream(torch.cuda.current_stream(param.device))
            with torch.cuda.stream(stream):
                for _ in range(2):
                    self.forward(x, label)                       # warm-up: weight packing, allocator pools
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                static_logits = self.forward(x, label)
        for i in range(shape[0]):
            for j in range(shape[1]):
                if use_graph:
                    graph.replay()
                    logits = static_logits
                else:
                    logits = self.forward(x, label)
                probs = torch.softmax(logits[:, :, i, j], -1)
                x[:, i, j].copy_(probs.multinomial(1).squeeze(-1))
        return x
