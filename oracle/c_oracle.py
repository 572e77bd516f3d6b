"""ctypes/numpy binding of oracle/vqvae_oracle.c  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under vqvae_amd/ does (the product path fails loudly when
the HIP library is missing, it never falls back to this).

All arrays are numpy, fp32 / int64 / int32, C-contiguous, NCHW at the
reference's module boundaries (models/vqvae.py:29-44).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvqvae_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vqvae_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libvqvae_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.vqo_aten_row_sum.restype = C.c_float
    return _lib


def _p(a, ty=C.c_float):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Weights(C.Structure):
    _fields_ = [("h_dim", C.c_int), ("res_h_dim", C.c_int), ("n_res_layers", C.c_int),
                ("n_embeddings", C.c_int), ("embedding_dim", C.c_int), ("in_ch", C.c_int),
                ("beta", C.c_float)] + [(n, C.POINTER(C.c_float)) for n in (
                    "enc0_w", "enc0_b", "enc2_w", "enc2_b", "enc4_w", "enc4_b",
                    "enc_res_w1", "enc_res_w2", "pre_w", "pre_b", "codebook",
                    "dec0_w", "dec0_b", "dec_res_w1", "dec_res_w2",
                    "dec2_w", "dec2_b", "dec4_w", "dec4_b")]


# state_dict key for each struct field
_SD_KEYS = {
    "enc0_w": "encoder.conv_stack.0.weight", "enc0_b": "encoder.conv_stack.0.bias",
    "enc2_w": "encoder.conv_stack.2.weight", "enc2_b": "encoder.conv_stack.2.bias",
    "enc4_w": "encoder.conv_stack.4.weight", "enc4_b": "encoder.conv_stack.4.bias",
    "enc_res_w1": "encoder.conv_stack.5.stack.0.res_block.1.weight",
    "enc_res_w2": "encoder.conv_stack.5.stack.0.res_block.3.weight",
    "pre_w": "pre_quantization_conv.weight", "pre_b": "pre_quantization_conv.bias",
    "codebook": "vector_quantization.embedding.weight",
    "dec0_w": "decoder.inverse_conv_stack.0.weight", "dec0_b": "decoder.inverse_conv_stack.0.bias",
    "dec_res_w1": "decoder.inverse_conv_stack.1.stack.0.res_block.1.weight",
    "dec_res_w2": "decoder.inverse_conv_stack.1.stack.0.res_block.3.weight",
    "dec2_w": "decoder.inverse_conv_stack.2.weight", "dec2_b": "decoder.inverse_conv_stack.2.bias",
    "dec4_w": "decoder.inverse_conv_stack.4.weight", "dec4_b": "decoder.inverse_conv_stack.4.bias",
}


class Model:
    """Holds a numpy copy of a VQVAE state_dict in the oracle's struct layout."""

    def __init__(self, state_dict, beta: float, n_res_layers: int):
        sd = {k: _f32(v.detach().cpu().numpy() if hasattr(v, "detach") else v)
              for k, v in state_dict.items()}
        self._keep = sd
        w = Weights()
        for f, k in _SD_KEYS.items():
            setattr(w, f, _p(sd[k]))
        w.h_dim = sd[_SD_KEYS["enc2_w"]].shape[0]
        w.res_h_dim = sd[_SD_KEYS["enc_res_w1"]].shape[0]
        w.n_res_layers = n_res_layers
        w.n_embeddings, w.embedding_dim = sd[_SD_KEYS["codebook"]].shape
        w.in_ch = sd[_SD_KEYS["enc0_w"]].shape[1]
        w.beta = beta
        self.w = w
        self.beta = beta

    @property
    def codebook(self):
        return self._keep[_SD_KEYS["codebook"]]

    def encode(self, x):
        x = _f32(x)
        B, _, H, W = x.shape
        z = np.empty((B, self.w.embedding_dim, H // 4, W // 4), np.float32)
        lib().vqo_encode(C.byref(self.w), _p(x), C.c_int64(B), H, W, _p(z))
        return z

    def decode(self, z_q):
        z_q = _f32(z_q)
        B, _, h, w = z_q.shape
        out = np.empty((B, self.w.in_ch, 4 * h, 4 * w), np.float32)
        lib().vqo_decode(C.byref(self.w), _p(z_q), C.c_int64(B), h, w, _p(out))
        return out

    def forward(self, x):
        """-> dict(loss, x_hat, perplexity, z_e, z_q, idx) (models/vqvae.py:29-44)."""
        x = _f32(x)
        B, _, H, W = x.shape
        D = self.w.embedding_dim
        x_hat = np.empty_like(x)
        z_e = np.empty((B, D, H // 4, W // 4), np.float32)
        z_q = np.empty_like(z_e)
        idx = np.empty((B * (H // 4) * (W // 4),), np.int64)
        loss, ppl = C.c_float(), C.c_float()
        lib().vqo_forward(C.byref(self.w), _p(x), C.c_int64(B), H, W, _p(x_hat),
                          C.byref(loss), C.byref(ppl), _p(z_e), _p(z_q), _p(idx, C.c_int64))
        return dict(loss=np.float32(loss.value), x_hat=x_hat, perplexity=np.float32(ppl.value),
                    z_e=z_e, z_q=z_q, idx=idx.reshape(-1, 1))


def row_sqnorm(x):
    x = _f32(x)
    rows, d = x.shape
    out = np.empty((rows,), np.float32)
    lib().vqo_row_sqnorm(_p(x), C.c_int64(rows), d, _p(out))
    return out


def vq_forward(z_nchw, codebook, beta, want_dist=False):
    """VectorQuantizer.forward (models/quantizer.py:29-76) ->
    dict(loss, z_q, perplexity, idx (N,1) int64, hist (K,) int32[, dist (N,K)])."""
    z = _f32(z_nchw)
    cb = _f32(codebook)
    B, D, H, W = z.shape
    K = cb.shape[0]
    assert cb.shape[1] == D
    N = B * H * W
    zq = np.empty_like(z)
    idx = np.empty((N,), np.int64)
    hist = np.empty((K,), np.int32)
    dist = np.empty((N, K), np.float32) if want_dist else None
    loss, ppl = C.c_float(), C.c_float()
    lib().vqo_vq_forward(_p(z), _p(cb), C.c_int64(B), D, H, W, K, C.c_float(beta), _p(zq),
                         _p(idx, C.c_int64), _p(hist, C.c_int32), C.byref(loss), C.byref(ppl),
                         _p(dist))
    out = dict(loss=np.float32(loss.value), z_q=zq, perplexity=np.float32(ppl.value),
               idx=idx.reshape(-1, 1), hist=hist)
    if want_dist:
        out["dist"] = dist
    return out


def vq_indices_rows(z_rows, codebook, threads=None, slab=8192):
    """min_encoding_indices (N,) int64 of row-major rows (N, D): vqo_vq_indices_rows -- the arithmetic of vq_forward, eight codes per
    AVX2 register -- with the row slabs spread over host threads (ctypes releases the GIL; rows are independent).  The checker for
    the full-size configs: 1.6 M rows x K = 1024 in about a second on the GPU box's host."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    z = _f32(z_rows)
    cb = _f32(codebook)
    N, D = z.shape
    K = cb.shape[0]
    assert cb.shape[1] == D
    idx = np.empty((N,), np.int64)
    fn = lib().vqo_vq_indices_rows
    if threads is None:
        threads = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 128))

    def work(lo):
        hi = min(N, lo + slab)
        rc = fn(_p(z[lo:hi]), _p(cb), C.c_int64(hi - lo), D, K, _p(idx[lo:hi], C.c_int64))
        if rc != 0:
            raise MemoryError("vqo_vq_indices_rows")
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(work, range(0, N, slab)))
    return idx


def onehot(idx, K):
    idx = np.ascontiguousarray(idx, np.int64).reshape(-1)
    out = np.empty((idx.shape[0], K), np.float32)
    lib().vqo_onehot(_p(idx, C.c_int64), C.c_int64(idx.shape[0]), K, _p(out))
    return out


def decode_indices(idx, codebook, B, H, W):
    idx = np.ascontiguousarray(idx, np.int64).reshape(-1)
    cb = _f32(codebook)
    D = cb.shape[1]
    out = np.empty((B, D, H, W), np.float32)
    lib().vqo_decode_indices(_p(idx, C.c_int64), _p(cb), C.c_int64(B), D, H, W, _p(out))
    return out


def conv2d(x, w, b, stride, pad, relu_in=False, relu_out=False):
    x, w = _f32(x), _f32(w)
    b = _f32(b) if b is not None else None
    B, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    y = np.empty((B, Cout, Ho, Wo), np.float32)
    lib().vqo_conv2d(_p(x), _p(w), _p(b), C.c_int64(B), Cin, H, W, Cout, kh, kw, stride, pad,
                     int(relu_in) | (int(relu_out) << 1), _p(y))
    return y


def conv_transpose2d(x, w, b, stride, pad, relu_in=False, relu_out=False):
    x, w = _f32(x), _f32(w)
    b = _f32(b) if b is not None else None
    B, Cin, H, W = x.shape
    _, Cout, kh, kw = w.shape
    Ho, Wo = (H - 1) * stride - 2 * pad + kh, (W - 1) * stride - 2 * pad + kw
    y = np.empty((B, Cout, Ho, Wo), np.float32)
    lib().vqo_conv_transpose2d(_p(x), _p(w), _p(b), C.c_int64(B), Cin, H, W, Cout, kh, kw,
                               stride, pad, int(relu_in) | (int(relu_out) << 1), _p(y))
    return y


def residual_stack(x, w1, w2, n_layers):
    x, w1, w2 = _f32(x), _f32(w1), _f32(w2)
    B, Cc, H, W = x.shape
    y = np.empty_like(x)
    lib().vqo_residual_stack(_p(x), _p(w1), _p(w2), C.c_int64(B), Cc, H, W, w1.shape[0],
                             n_layers, _p(y))
    return y
