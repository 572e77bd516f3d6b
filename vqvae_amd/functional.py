"""torch-tensor front end of the C ABI: allocates outputs/workspaces with torch's
caching allocator, passes raw device pointers and the current HIP stream to
libvqvae_hip.so.  PyTorch is plumbing here (memory + streams); every kernel that
runs is ours.  No CPU path: CPU tensors are rejected.
"""
from __future__ import annotations

import torch

from . import _lib

VQ_ROWMAJOR = 0x1
VQ_CODEBOOK_PREPARED = 0x2
VQ_EXACT_SWEEP = 0x4
VQ_BF16_FILTER = 0x8
VQ_UNFUSED = 0x40
VQ_UNITS64_8WAVES = 0x100
VQ_UNITS32_16WAVES = 0x200
# whole-path product scheme (vqvae_forward_f32 / vqvae_encoder_ex_f32 / vqvae_decoder_ex_f32)
FWD_CONV_BF16_SPLIT = 0x1000
FWD_CONV_EXACT_FP32 = 0x2000
FWD_DEBUG_ZE = 0x4000            # tests: the fused encoder+quantizer kernel also writes its z_e (include/vqvae_hip.h)
VQ_UNITS32_8WAVES = 0x400


def _stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _check_dev(name, t, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.VqvaeHipError(f"{name} must be a CUDA(HIP) tensor: the MI355X path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype} (the reference path is fp32 only)")


def vq_workspace(K: int, D: int, device) -> torch.Tensor:
    n = _lib.load().vqvae_vq_workspace_bytes(0, K, D)
    if n == 0:
        raise _lib.VqvaeHipError(f"VectorQuantizer shape K={K}, D={D} not supported by the gfx950 kernels "
                                 "(D <= 256, K <= 16384)")
    return torch.empty(n, dtype=torch.uint8, device=device)


def vq_forward(z_e: torch.Tensor, codebook: torch.Tensor, beta: float, *, rowmajor: bool = False,
               workspace: torch.Tensor | None = None, prepared: bool = False, want_zq: bool = True,
               exact_sweep: bool = False, bf16_filter: bool = False, form: int = 0):
    """Fused VectorQuantizer forward (models/quantizer.py:29-76).

    z_e: (B,D,H,W) contiguous, or (B,H,W,D) contiguous when rowmajor.
    Returns (loss 0-dim, z_q like z_e or None, perplexity 0-dim, idx (N,1) int64, hist (K,) int32).
    exact_sweep=True forces the exhaustive fp32-MFMA kernel, bf16_filter=True round 1's two-sweep bf16 filter
    kernel, instead of the default (single-sweep fp16 screen with the stream tracker: D=64 rows, row-major or NCHW maps of
    32 k pixels); all three produce identical bits, the flags exist for testing and A/B timing.  form: 8 / 16 forces the
    stream-tracker kernel's launch form (64-row units on eight waves / 32-row units on sixteen waves per CU; 0 = by row count).
    """
    _check_dev("z_e", z_e)
    _check_dev("codebook", codebook)
    if z_e.dim() != 4:
        raise ValueError("z_e must be 4-D")
    if rowmajor:
        B, H, W, D = z_e.shape
    else:
        B, D, H, W = z_e.shape
    K, Dc = codebook.shape
    if Dc != D:
        # the reference silently mis-reshapes here (view(-1, e_dim), quantizer.py:46); be strict
        raise ValueError(f"channel dim {D} != embedding dim {Dc}")
    z_e = z_e.contiguous()
    codebook = codebook.contiguous()
    dev = z_e.device
    with torch.cuda.device(dev):
        if workspace is None:
            workspace = vq_workspace(K, D, dev)
            prepared = False
        N = B * H * W
        z_q = torch.empty_like(z_e) if want_zq else None
        idx = torch.empty((N, 1), dtype=torch.int64, device=dev)
        hist = torch.empty((K,), dtype=torch.int32, device=dev)
        scal = torch.empty((2,), dtype=torch.float32, device=dev)
        flags = (VQ_ROWMAJOR if rowmajor else 0) | (VQ_CODEBOOK_PREPARED if prepared else 0) | \
            (VQ_EXACT_SWEEP if exact_sweep else 0) | (VQ_BF16_FILTER if bf16_filter else 0) | \
            (VQ_UNITS32_16WAVES if form == 16 else (VQ_UNITS64_8WAVES if form == 8 else 0))
        rc = _lib.load().vqvae_vq_forward_f32(
            z_e.data_ptr(), codebook.data_ptr(), B, D, H, W, K, float(beta), flags,
            z_q.data_ptr() if want_zq else None, idx.data_ptr(), hist.data_ptr(),
            scal.data_ptr(), scal.data_ptr() + 4, workspace.data_ptr(), workspace.numel(),
            _stream_ptr(z_e))
        _lib.check(rc)
    return scal[0], z_q, scal[1], idx, hist


def vq_onehot(idx: torch.Tensor, K: int) -> torch.Tensor:
    """min_encodings (N,K) fp32 (models/quantizer.py:55-57)."""
    _check_dev("idx", idx, torch.int64)
    idx = idx.contiguous()
    N = idx.numel()
    out = torch.empty((N, K), dtype=torch.float32, device=idx.device)
    with torch.cuda.device(idx.device):
        _lib.check(_lib.load().vqvae_vq_onehot_f32(idx.data_ptr(), N, K, out.data_ptr(), _stream_ptr(idx)))
    return out


def vq_decode_indices(idx: torch.Tensor, codebook: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """indices -> z_q (B,D,H,W) (visualization.ipynb:358-365)."""
    _check_dev("idx", idx, torch.int64)
    _check_dev("codebook", codebook)
    idx = idx.contiguous()
    codebook = codebook.contiguous()
    K, D = codebook.shape
    if idx.numel() != B * H * W:
        raise ValueError("idx must hold B*H*W indices")
    # the reference's embedding lookup raises on an out-of-range index; the kernel cannot (it writes NaN), so the
    # check is done here -- one host sync, skipped while a graph is being captured
    if not torch.cuda.is_current_stream_capturing():
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= K:
            raise IndexError(f"index out of range in decode_indices: [{lo}, {hi}] not within [0, {K})")
    out = torch.empty((B, D, H, W), dtype=torch.float32, device=idx.device)
    with torch.cuda.device(idx.device):
        _lib.check(_lib.load().vqvae_vq_decode_indices_f32(idx.data_ptr(), codebook.data_ptr(), B, D, H, W, K,
                                                           out.data_ptr(), _stream_ptr(idx)))
    return out
