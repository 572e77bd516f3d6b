"""Host-side mirror of the reference's module interface for the forward hot path.

Same class names, constructor signatures, attribute names, parameter names/shapes
and `state_dict` layout as the reference, so checkpoints written by
the reference (`utils.py:109-113`) load unchanged and callers (`main.py:74`,
`visualization.ipynb:84-90`) keep working:

    VQVAE(h_dim, res_h_dim, n_res_layers, n_embeddings, embedding_dim, beta,
          save_img_embedding_map=False).forward(x) -> (embedding_loss, x_hat, perplexity)
                                                              (models/vqvae.py:11-12,29-44)
    VectorQuantizer(n_e, e_dim, beta).forward(z)
          -> (loss, z_q, perplexity, min_encodings, min_encoding_indices)   (models/quantizer.py:20,29-76)
    Encoder(in_dim, h_dim, n_res_layers, res_h_dim)                          (models/encoder.py:24)
    Decoder(in_dim, h_dim, n_res_layers, res_h_dim)                          (models/decoder.py:22)
    ResidualLayer(in_dim, h_dim, res_h_dim), ResidualStack(in_dim, h_dim, res_h_dim, n_res_layers)
                                                                             (models/residual.py:16,41)

The modules only HOLD parameters; the arithmetic is libvqvae_hip.so, on CUDA(HIP)
fp32 tensors only -- there is no CPU fallback.  `VQVAE.forward` under autograd (main.py:74-79) runs every
layer forward and backward on the HIP kernels (autograd_conv.py, training.VQStraightThrough; SURVEY.md 8f
row 2).  The sub-modules called on their own (`Encoder`, `Decoder`, `ResidualStack`) are forward-only on the
HIP backend and raise while a graph is being recorded.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _cache
from . import functional as F_hip
from ._lib import VqvaeHipError


def _records_graph(*tensors):
    """Is a graph being recorded through the HIP conv path?  (round 5: the sub-modules then run autograd_conv's Functions --
    forward AND backward on the HIP kernels -- as VQVAE.forward does; models/encoder.py:42-43, decoder.py:38-39,
    residual.py:47-51 are differentiable upstream, and so are they here)"""
    from . import conv as C_hip
    return C_hip.get_conv_backend() == "hip" and torch.is_grad_enabled() and any(t.requires_grad for t in tensors)


def _require_forward_only(*tensors):
    """The whole-path / fused forward entries keep no activations: refuse to run them while a graph is recorded."""
    if _records_graph(*tensors):
        raise VqvaeHipError(
            "this entry point is forward-only: call it under torch.no_grad() (or model.requires_grad_(False)); "
            "VQVAE.forward, Encoder / Decoder / ResidualStack.forward record a graph on the HIP kernels by themselves.  "
            "There is no silent fallback")


def _need_hip_f32(x, who):
    if not x.is_cuda or x.dtype != torch.float32:
        raise VqvaeHipError(f"{who} needs a CUDA(HIP) fp32 input: there is no CPU path")


class LazyOneHot:
    """`min_encodings` of VectorQuantizer.forward (models/quantizer.py:55-57): the (N, K) fp32 one-hot -- 512 MiB at BASELINE config
    3's size, 128 GiB at config 5's -- which neither caller of the reference reads (models/vqvae.py:34 and visualization.ipynb:87
    both discard it).  SURVEY.md 8b: materialise lazily.  This stands in the return tuple, knows its shape / dtype / device, and
    becomes the real tensor (vqvae_vq_onehot_f32, once) the moment anything is done with it: attribute access, indexing, arithmetic,
    or being passed to a torch function.  `VectorQuantizer.LAZY_MIN_ENCODINGS = False` returns the tensor itself as before."""

    def __init__(self, idx, n_e):
        self._idx, self._n_e, self._t = idx, n_e, None

    def materialize(self):
        if self._t is None:
            self._t = F_hip.vq_onehot(self._idx, self._n_e)
        return self._t

    @property
    def shape(self):
        return torch.Size((self._idx.shape[0], self._n_e))

    @property
    def dtype(self):
        return torch.float32

    @property
    def device(self):
        return self._idx.device

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 2

    def numel(self):
        return self._idx.shape[0] * self._n_e

    def __len__(self):
        return self._idx.shape[0]

    def __getattr__(self, name):                      # anything else: the tensor's
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def real(a):
            if isinstance(a, LazyOneHot):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(v) for v in a)
            return a
        return func(*real(args), **{k: real(v) for k, v in (kwargs or {}).items()})

    def __repr__(self):
        return f"LazyOneHot(shape={tuple(self.shape)}, materialized={self._t is not None})"


def _delegate(name):
    def op(self, *a, **k):
        return getattr(self.materialize(), name)(*[v.materialize() if isinstance(v, LazyOneHot) else v for v in a], **k)
    op.__name__ = name
    return op


for _n in ("__getitem__", "__iter__", "__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__matmul__",
           "__rmatmul__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__neg__", "__bool__", "__float__", "__int__",
           "__array__"):
    setattr(LazyOneHot, _n, _delegate(_n))
LazyOneHot.__hash__ = object.__hash__


class VectorQuantizer(nn.Module):
    """Discretisation bottleneck; mirrors models/quantizer.py:10-76."""

    LAZY_MIN_ENCODINGS = True          # forward()'s fourth output: LazyOneHot (the (N, K) one-hot on first use) or the tensor itself

    def __init__(self, n_e, e_dim, beta):
        super().__init__()
        self.n_e = n_e
        self.e_dim = e_dim
        self.beta = beta
        self.embedding = nn.Embedding(self.n_e, self.e_dim)
        self.embedding.weight.data.uniform_(-1.0 / self.n_e, 1.0 / self.n_e)   # quantizer.py:27

    def invalidate(self):
        """Forget the prepared codebook image.  The cache is keyed on (data_ptr, _version, device), which in-place ops
        and optimizers bump; writes through `.data` (`weight.data.copy_()`, EMA updates) do NOT -- call this after
        such a write.  `load_state_dict` and `.to()/.cuda()` call it for you."""
        for slot in _cache.side(self).get("ws", {}).values():
            slot[1] = None

    def _workspace(self):
        """-> (workspace, prepared, key, slot): workspace holding the codebook's LDS images and the per-call scratch,
        ONE PER (device, stream) -- two forwards of one model on different streams must not share scratch -- re-prepared
        only when the embedding tensor changes.  The caller stores `key` in slot[1] once the launch has succeeded."""
        w = self.embedding.weight
        key = (w.data_ptr(), w._version, w.device)
        table = _cache.side(self).setdefault("ws", {})
        skey = (str(w.device), torch.cuda.current_stream(w.device).cuda_stream if w.is_cuda else 0)
        slot = _cache.lru_get(table, skey)
        if slot is None:
            slot = [F_hip.vq_workspace(self.n_e, self.e_dim, w.device), None]
            _cache.lru_put(table, skey, slot, 8)                 # streams come and go: the least recently used slot leaves
        return slot[0], slot[1] == key, key, slot

    def quantize(self, z, *, rowmajor=False, want_zq=True):
        """-> (loss, z_q, perplexity, min_encoding_indices, hist); no one-hot."""
        w = self.embedding.weight
        ws, prepared, key, slot = self._workspace()
        if not prepared:
            slot[1] = None                                       # a failed launch must not leave a stale "prepared" image
        if torch.is_grad_enabled() and (z.requires_grad or w.requires_grad):
            from .training import VQStraightThrough          # HIP forward + HIP backward
            out = VQStraightThrough.apply(z, w, self.beta, rowmajor, ws, prepared)
        else:
            out = F_hip.vq_forward(z, w.detach(), self.beta, rowmajor=rowmajor, workspace=ws,
                                   prepared=prepared, want_zq=want_zq)
        slot[1] = key
        return out

    def _apply(self, fn, *args, **kwargs):
        self.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def forward(self, z):
        loss, z_q, perplexity, idx, _ = self.quantize(z)
        min_encodings = LazyOneHot(idx, self.n_e) if self.LAZY_MIN_ENCODINGS else F_hip.vq_onehot(idx, self.n_e)   # quantizer.py:55-57
        return loss, z_q, perplexity, min_encodings, idx


class ResidualLayer(nn.Module):
    """Parameter holder mirroring models/residual.py:8-29."""

    def __init__(self, in_dim, h_dim, res_h_dim):
        super().__init__()
        self.res_block = nn.Sequential(
            nn.ReLU(True),
            nn.Conv2d(in_dim, res_h_dim, kernel_size=3, stride=1, padding=1, bias=False),
            nn.ReLU(True),
            nn.Conv2d(res_h_dim, h_dim, kernel_size=1, stride=1, bias=False),
        )

    def forward(self, x):
        return ResidualStack._run(x, [self], final_relu=False, mutate_input=True)


class ResidualStack(nn.Module):
    """n_res_layers aliases of ONE ResidualLayer (models/residual.py:41-51): the
    state_dict lists stack.0..n-1 keys that all share storage, exactly as upstream."""

    def __init__(self, in_dim, h_dim, res_h_dim, n_res_layers):
        super().__init__()
        self.n_res_layers = n_res_layers
        self.stack = nn.ModuleList([ResidualLayer(in_dim, h_dim, res_h_dim)] * n_res_layers)

    @staticmethod
    def _run(x, layers, final_relu, mutate_input):
        from . import conv as C_hip
        if _records_graph(x, *[p for l in layers for p in l.parameters()]):
            # under autograd the caller's tensor is NOT rewritten to relu(x) (upstream's in-place nn.ReLU does that through
            # autograd's version counter; an out-of-graph write here would corrupt other consumers' gradients): use the return value
            from . import autograd_conv as A_hip
            _need_hip_f32(x, "ResidualStack.forward")
            return A_hip.residual_stack_forward_train(x, layers, final_relu)
        y = C_hip.residual_stack_nchw(x, layers, final_relu=final_relu)
        if mutate_input:
            # nn.ReLU(True) upstream rewrites the caller's tensor to relu(x) (residual.py:19)
            x.relu_()
        return y

    def forward(self, x):
        return ResidualStack._run(x, list(self.stack), final_relu=True, mutate_input=True)


class Encoder(nn.Module):
    """q_theta(z|x) parameter holder mirroring models/encoder.py:24-43."""

    def __init__(self, in_dim, h_dim, n_res_layers, res_h_dim):
        super().__init__()
        kernel, stride = 4, 2
        self.conv_stack = nn.Sequential(
            nn.Conv2d(in_dim, h_dim // 2, kernel_size=kernel, stride=stride, padding=1),
            nn.ReLU(),
            nn.Conv2d(h_dim // 2, h_dim, kernel_size=kernel, stride=stride, padding=1),
            nn.ReLU(),
            nn.Conv2d(h_dim, h_dim, kernel_size=kernel - 1, stride=stride - 1, padding=1),
            ResidualStack(h_dim, h_dim, res_h_dim, n_res_layers),
        )

    def forward(self, x):
        from . import conv as C_hip
        if _records_graph(x, *self.parameters()):
            from . import autograd_conv as A_hip
            _need_hip_f32(x, "Encoder.forward")
            return A_hip.encoder_stack_forward_train(self, x)
        return C_hip.encoder_forward(self, x, pre_quant=None)


class Decoder(nn.Module):
    """p_phi(x|z) parameter holder mirroring models/decoder.py:22-39."""

    def __init__(self, in_dim, h_dim, n_res_layers, res_h_dim):
        super().__init__()
        kernel, stride = 4, 2
        self.inverse_conv_stack = nn.Sequential(
            nn.ConvTranspose2d(in_dim, h_dim, kernel_size=kernel - 1, stride=stride - 1, padding=1),
            ResidualStack(h_dim, h_dim, res_h_dim, n_res_layers),
            nn.ConvTranspose2d(h_dim, h_dim // 2, kernel_size=kernel, stride=stride, padding=1),
            nn.ReLU(),
            nn.ConvTranspose2d(h_dim // 2, 3, kernel_size=kernel, stride=stride, padding=1),
        )

    def forward(self, x):
        from . import conv as C_hip
        if _records_graph(x, *self.parameters()):
            from . import autograd_conv as A_hip
            _need_hip_f32(x, "Decoder.forward")
            return A_hip.decoder_forward_train(self, x.permute(0, 2, 3, 1).contiguous())
        return C_hip.decoder_forward(self, x, rowmajor_in=False)


def _invalidate_after_load(module, incompatible):
    """load_state_dict post-hook (a module-level function: a local lambda here made the model unpicklable)."""
    module.invalidate_caches()


class VQVAE(nn.Module):
    """Mirrors models/vqvae.py:10-44."""

    def __init__(self, h_dim, res_h_dim, n_res_layers, n_embeddings, embedding_dim, beta,
                 save_img_embedding_map=False):
        super().__init__()
        self.encoder = Encoder(3, h_dim, n_res_layers, res_h_dim)
        self.pre_quantization_conv = nn.Conv2d(h_dim, embedding_dim, kernel_size=1, stride=1)
        self.vector_quantization = VectorQuantizer(n_embeddings, embedding_dim, beta)
        self.decoder = Decoder(embedding_dim, h_dim, n_res_layers, res_h_dim)
        if save_img_embedding_map:
            self.img_to_embedding_map = {i: [] for i in range(n_embeddings)}
        else:
            self.img_to_embedding_map = None
        self.register_load_state_dict_post_hook(_invalidate_after_load)

    def invalidate_caches(self):
        """Drop the prepared codebook image and every packed-weight image (they are keyed on the parameters'
        data_ptr / _version, which writes through `.data` do not change; `load_state_dict` calls this)."""
        from . import conv_hip
        _cache.drop(self, "c_weights")
        for mod in self.modules():
            conv_hip.invalidate(mod)
            if isinstance(mod, VectorQuantizer):
                mod.invalidate()

    # forward / encode / decode_indices follow vqvae_weights_range_check_f32's recommendation for the checkpoint (scheme_hint()) unless
    # the caller names a scheme.  False: never check (no host sync per weight version -- an eval forward after every optimizer step
    # pays one otherwise), always the default two-term fp16 products unless a scheme is named.
    AUTO_SCHEME = True

    # ---- whole-path C ABI (include/vqvae_hip.h: vqvae_forward_f32) ------------------------------------------------
    _RAW = {"enc0_w": "encoder.conv_stack.0.weight", "enc0_b": "encoder.conv_stack.0.bias",
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
            "dec4_w": "decoder.inverse_conv_stack.4.weight", "dec4_b": "decoder.inverse_conv_stack.4.bias"}

    def _c_weights(self):
        """-> (VqvaeWeights, keep-alive tensors): every layer packed once per weight version into ONE buffer by
        vqvae_weights_pack_f32; rebuilt when any parameter's (data_ptr, _version) changes or invalidate_caches() ran."""
        from . import _lib
        params = dict(self.named_parameters(remove_duplicate=False))
        n_res = self.encoder.conv_stack[5].n_res_layers
        tensors = {}
        for f, k in self._RAW.items():
            if k not in params:                       # n_res_layers == 0: no residual weights; the stack is skipped
                tensors[f] = params["encoder.conv_stack.4.weight"]
            else:
                tensors[f] = params[k]
        key = tuple((t.data_ptr(), t._version) for t in tensors.values()) + (str(tensors["enc0_w"].device),)
        hit = _cache.side(self).get("c_weights")
        if hit is not None and hit[0] == key:
            _cache.wait_ready(hit[3], tensors["enc0_w"].device)          # packed on another stream a moment ago?
            return hit[1], hit[2]
        L = _lib.load()
        w0 = tensors["enc0_w"]
        dims = _lib.VqvaeDims(self.encoder.conv_stack[4].weight.shape[0], tensors["enc_res_w1"].shape[0] if n_res else 1,
                              n_res, self.vector_quantization.n_e, self.vector_quantization.e_dim, w0.shape[1],
                              float(self.vector_quantization.beta))
        nbytes = L.vqvae_weights_packed_bytes(dims)
        if nbytes == 0:
            raise VqvaeHipError("model dimensions not supported by the gfx950 whole-path entry points")
        keep = {f: t.detach().contiguous() for f, t in tensors.items()}
        raw = _lib.VqvaeRawWeights(**{f: t.data_ptr() for f, t in keep.items()})
        packed = torch.empty(nbytes, dtype=torch.uint8, device=w0.device)
        cw = _lib.VqvaeWeights()
        with torch.cuda.device(w0.device):
            _lib.check(L.vqvae_weights_pack_f32(dims, raw, packed.data_ptr(), nbytes, cw,
                                                torch.cuda.current_stream(w0.device).cuda_stream))
            # which product scheme this checkpoint calls for (vqvae_weights_range_check_f32: once per weight version, one sync).
            # Not while a stream capture is recording (the check synchronises): the previous version's hint stands, or none.
            import ctypes
            rec = ctypes.c_int(0)
            spreads = (ctypes.c_float * 11)()
            hint = None
            if self.AUTO_SCHEME and not torch.cuda.is_current_stream_capturing():
                scratch = torch.empty(16, dtype=torch.float32, device=w0.device)
                _lib.check(L.vqvae_weights_range_check_f32(dims, raw, spreads, ctypes.byref(rec), scratch.data_ptr(), 64,
                                                           torch.cuda.current_stream(w0.device).cuda_stream))
                hint = (int(rec.value), [float(v) for v in spreads])
            elif self.AUTO_SCHEME:
                hint = _cache.side(self).get("c_scheme_hint")
                import warnings
                warnings.warn("VQVAE: new weights met inside a stream capture -- the range check (a host sync) is skipped and "
                              + ("the previous weights' product scheme is kept" if hint else "the default two-term fp16 scheme is used")
                              + "; run one forward outside the capture first, or name a scheme (fwd_flags)", RuntimeWarning, stacklevel=3)
            ready = _cache.mark_ready(w0.device)
        if hint is None:
            hint = (0, [0.0] * 11)
        old = _cache.side(self).get("c_scheme_hint")
        if hint[0] and (old is None or old[0] != hint[0]):
            # the whole path changes kernels, numerics and speed with this: say so once per flip, never silently
            import warnings
            warnings.warn(f"VQVAE: this checkpoint's weights spread over {max(hint[1]):.1f} binades of input-channel magnitude in one layer "
                          "(limit 10): forward / encode / decode_indices run the three-term bf16 scheme (VQVAE_FWD_CONV_BF16_SPLIT, about "
                          "half the speed of the default two-term fp16 products).  Name a scheme (fwd_flags=...) or set "
                          "VQVAE.AUTO_SCHEME = False to decide yourself", RuntimeWarning, stacklevel=3)
        _cache.side(self)["c_weights"] = (key, cw, (keep, packed), ready)
        _cache.side(self)["c_scheme_hint"] = hint
        return cw, (keep, packed)

    def scheme_hint(self):
        """-> (flags, per-layer input-channel spread in binades): FWD_CONV_BF16_SPLIT when this checkpoint's weights put the default
        two-term fp16 products outside their range (include/vqvae_hip.h, vqvae_weights_range_check_f32), else 0.  forward / encode /
        decode_indices apply it unless the caller names a scheme."""
        self._c_weights()
        return _cache.side(self)["c_scheme_hint"]

    # The step can run as n parts on n side streams (vqvae_forward_begin / part / end; _forward_c(x, parts=n)): the kernels of
    # different parts fill each other's ramp-up and tail.  NOT the default: measured -2 % per step on one MI355X box and +9 % on
    # another (profiles/r03_notes.txt section 11), so FORWARD_PARTS stays 1 = one vqvae_forward_f32 call on the current stream.
    FORWARD_PARTS = 1
    FORWARD_PARTS_MIN_BATCH = 2048

    def _forward_parts(self, L, cw, x, B, H, W, flags, x_hat, scal, idx, ws, nws, vws, dev, stream, parts):
        """The step in parts on side streams; False = not applicable here (the caller makes the single call).  Outputs are
        bit-identical to the single call; the current stream waits for the side streams before this returns."""
        from . import _lib
        n = self.FORWARD_PARTS if parts is None else parts
        if n < 2 or B < (self.FORWARD_PARTS_MIN_BATCH if parts is None else 128) or (H, W) != (32, 32):
            return False
        rc = L.vqvae_forward_begin_f32(cw, B, H, W, flags, ws.data_ptr(), nws, vws.data_ptr(), vws.numel(), stream)
        if rc == -3:                                   # VQVAE_ERR_UNSUPPORTED: another architecture / codebook shape / quantizer flags
            return False
        _lib.check(rc)
        pool = _cache.side(self).setdefault("part_streams", {})
        streams = pool.get(str(dev))
        if streams is None or len(streams) < n:
            streams = pool[str(dev)] = [torch.cuda.Stream(device=dev) for _ in range(n)]
        cur = torch.cuda.current_stream(dev)
        per = -(-B // n)
        per = -(-per // 64) * 64                        # whole 64-image blocks per part
        b0 = 0
        used = []
        try:
            for s in streams[:n]:
                if b0 >= B:
                    break
                bc = min(per, B - b0)
                s.wait_stream(cur)
                used.append(s)                                  # (before the launch: a failing part may have enqueued kernels already)
                _lib.check(L.vqvae_forward_part_f32(cw, x.data_ptr(), B, b0, bc, H, W, flags, x_hat.data_ptr(),
                                                    idx.data_ptr() if idx is not None else None, ws.data_ptr(), nws, vws.data_ptr(),
                                                    vws.numel(), s.cuda_stream))
                b0 += bc
        except BaseException:
            L.vqvae_forward_abort_f32(ws.data_ptr())              # the step will not be finished: drop its host-side record
            raise
        finally:
            # ALWAYS: the caller's stream must not free or reuse x_hat / ws / idx while a side stream still writes them, also
            # when a part raised
            for s in used:
                cur.wait_stream(s)
        _lib.check(L.vqvae_forward_end_f32(cw, B, H, W, scal.data_ptr(), scal.data_ptr() + 4, ws.data_ptr(), nws, stream))
        return True

    def _forward_c(self, x, want_idx=False, vq_flags=0, parts=None, fwd_flags=None):
        """VQVAE.forward as ONE call into libvqvae_hip.so (vqvae_forward_f32).  vq_flags: extra quantizer flags for tests and
        A/B runs (functional.VQ_UNFUSED: the quantizer as its own launch where the encoder's last kernel would quantize).
        parts: None = the default policy (FORWARD_PARTS side streams for large batches), 1 = always the single call, n = n parts.
        fwd_flags: functional.FWD_CONV_BF16_SPLIT / FWD_CONV_EXACT_FP32 = the whole path on the three-term bf16 / exact-fp32 MFMA
        kernels instead of the default two-term fp16 products; 0 = the two-term fp16 products whatever the weights; None (default) =
        what vqvae_weights_range_check_f32 recommends for this checkpoint (scheme_hint())."""
        from . import _lib
        L = _lib.load()
        x = x.contiguous()
        B, Cin, H, W = x.shape
        cw, _keep = self._c_weights()
        if fwd_flags is None:
            fwd_flags = _cache.side(self)["c_scheme_hint"][0]
        if Cin != cw.dims.in_ch or H % 4 or W % 4:
            raise ValueError(f"expected (B, {cw.dims.in_ch}, 4k, 4m) images, got {tuple(x.shape)}")
        dev = x.device
        with torch.cuda.device(dev):
            nws = L.vqvae_workspace_bytes(cw.dims, B, H, W)
            if nws == 0:
                raise VqvaeHipError(f"shape {tuple(x.shape)} not supported by the gfx950 whole-path entry points")
            # one activation workspace per (shape, device, stream), reused: forwards on different streams never share one
            stream = torch.cuda.current_stream(dev).cuda_stream
            # (keyed on (device, stream) only: ONE buffer per stream, grown to the largest batch seen -- a model that alternates
            # a training and an evaluation batch size keeps one workspace, not one per size)
            wkey = (str(dev), stream)
            table = _cache.side(self).setdefault("c_ws", {})
            ws = _cache.lru_get(table, wkey)
            if ws is None or ws.numel() < nws:
                ws = torch.empty(nws, dtype=torch.uint8, device=dev)
                _cache.lru_put(table, wkey, ws, 4)
            nws = ws.numel()
            vq = self.vector_quantization
            vws, prepared, key, slot = vq._workspace()
            if not prepared:
                slot[1] = None
            x_hat = torch.empty_like(x)
            scal = torch.empty(2, dtype=torch.float32, device=dev)
            idx = torch.empty((B * (H // 4) * (W // 4), 1), dtype=torch.int64, device=dev) if want_idx else None
            flags = (F_hip.VQ_CODEBOOK_PREPARED if prepared else 0) | vq_flags | fwd_flags
            if not self._forward_parts(L, cw, x, B, H, W, flags, x_hat, scal, idx, ws, nws, vws, dev, stream, parts):
                _lib.check(L.vqvae_forward_f32(cw, x.data_ptr(), B, H, W, flags, x_hat.data_ptr(), scal.data_ptr(), scal.data_ptr() + 4,
                                               idx.data_ptr() if want_idx else None, ws.data_ptr(), nws, vws.data_ptr(), vws.numel(),
                                               stream))
            slot[1] = key
        return (scal[0], x_hat, scal[1]) + ((idx,) if want_idx else ())

    def forward(self, x, verbose=False):
        from . import conv as C_hip
        if (C_hip.get_conv_backend() == "hip" and torch.is_grad_enabled()
                and any(p.requires_grad for p in self.parameters())):
            # training (main.py:74-79): every layer forward AND backward on the HIP kernels
            from . import autograd_conv as A_hip
            if not x.is_cuda or x.dtype != torch.float32:
                raise VqvaeHipError("VQVAE.forward needs a CUDA(HIP) fp32 input: there is no CPU path")
            z_e = A_hip.encoder_forward_train(self.encoder, x, self.pre_quantization_conv)
            embedding_loss, z_q, perplexity, _, _ = self.vector_quantization.quantize(z_e, rowmajor=True)
            x_hat = A_hip.decoder_forward_train(self.decoder, z_q)
            return embedding_loss, x_hat, perplexity
        _require_forward_only(x, *self.parameters())
        if C_hip.get_conv_backend() == "hip" and not verbose and x.is_cuda and x.dtype == torch.float32:
            return self._forward_c(x)                 # one ctypes call: vqvae_forward_f32
        # encoder + 1x1 pre-quantisation conv, activations kept row-major (B,H,W,C)
        z_e = C_hip.encoder_forward(self.encoder, x, pre_quant=self.pre_quantization_conv)
        embedding_loss, z_q, perplexity, _, _ = self.vector_quantization.quantize(z_e, rowmajor=True)
        x_hat = C_hip.decoder_forward(self.decoder, z_q, rowmajor_in=True)
        if verbose:                                                    # models/vqvae.py:38-42
            print('original data shape:', x.shape)
            print('encoded data shape:', torch.Size((z_e.shape[0], z_e.shape[3], z_e.shape[1], z_e.shape[2])))
            print('recon data shape:', x_hat.shape)
            assert False
        return embedding_loss, x_hat, perplexity

    # ---- "next" rows of SURVEY.md 8f-1: the index wire format -------------------
    def _c_workspace(self, L, cw, B, H, W, dev):
        """(workspace tensor, stream handle) of the whole-path C entry points for this device's current stream (see _forward_c)."""
        nws = L.vqvae_workspace_bytes(cw.dims, B, H, W)
        if nws == 0:
            raise VqvaeHipError(f"shape ({B}, {cw.dims.in_ch}, {H}, {W}) not supported by the gfx950 whole-path entry points")
        stream = torch.cuda.current_stream(dev).cuda_stream
        table = _cache.side(self).setdefault("c_ws", {})
        ws = _cache.lru_get(table, (str(dev), stream))
        if ws is None or ws.numel() < nws:
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)
            _cache.lru_put(table, (str(dev), stream), ws, 4)
        return ws, stream

    @torch.no_grad()
    def encode(self, x, vq_flags=0):
        """x -> min_encoding_indices (N,1) int64 (README.md:56; notebook encode_data) as ONE call into libvqvae_hip.so
        (vqvae_encode_f32): on the default shapes the encoder's last kernel quantizes its own z_e and only the indices are
        written -- no z_e, no z_q."""
        from . import _lib, conv as C_hip
        if C_hip.get_conv_backend() != "hip" or not x.is_cuda or x.dtype != torch.float32:
            z_e = C_hip.encoder_forward(self.encoder, x, pre_quant=self.pre_quantization_conv)
            _, _, _, idx, _ = self.vector_quantization.quantize(z_e, rowmajor=True, want_zq=False)
            return idx
        L = _lib.load()
        x = x.contiguous()
        B, Cin, H, W = x.shape
        cw, _keep = self._c_weights()
        if Cin != cw.dims.in_ch or H % 4 or W % 4:
            raise ValueError(f"expected (B, {cw.dims.in_ch}, 4k, 4m) images, got {tuple(x.shape)}")
        dev = x.device
        with torch.cuda.device(dev):
            ws, stream = self._c_workspace(L, cw, B, H, W, dev)
            vq = self.vector_quantization
            vws, prepared, key, slot = vq._workspace()
            if not prepared:
                slot[1] = None
            idx = torch.empty((B * (H // 4) * (W // 4), 1), dtype=torch.int64, device=dev)
            _lib.check(L.vqvae_encode_f32(cw, x.data_ptr(), B, H, W, (F_hip.VQ_CODEBOOK_PREPARED if prepared else 0) | vq_flags |
                                          (_cache.side(self)["c_scheme_hint"][0] if not (vq_flags & 0x3000) else 0),
                                          idx.data_ptr(), ws.data_ptr(), ws.numel(), vws.data_ptr(), vws.numel(), stream))
            slot[1] = key
        return idx

    @torch.no_grad()
    def decode_indices(self, idx, B, H, W, fwd_flags=None, validate=True):
        """indices -> x_hat (visualization.ipynb:358-365 generate_samples) as ONE call (vqvae_decode_f32): on the default shapes
        the decoder's first kernel takes every latent pixel's row straight from the codebook -- z_q is never written.
        H, W: the LATENT map's size.  Indices outside [0, K) raise, as the reference's one-hot scatter does -- that check reads
        idx.min() / idx.max() on the host (one sync per call); validate=False skips it for indices that come from encode(): the C
        entry never reads outside the codebook either way, a bad index shows as NaN pixels of its image (include/vqvae_hip.h)."""
        from . import _lib, conv as C_hip
        K = self.vector_quantization.n_e
        if idx.numel() != B * H * W:
            raise ValueError(f"expected {B * H * W} indices, got {idx.numel()}")
        if C_hip.get_conv_backend() != "hip" or not idx.is_cuda:
            z_q = F_hip.vq_decode_indices(idx, self.vector_quantization.embedding.weight.detach(), B, H, W)
            return self.decoder(z_q)
        idx = idx.contiguous().view(-1).to(torch.int64)
        if validate and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= K):
            raise IndexError(f"code index out of range [0, {K})")
        L = _lib.load()
        cw, _keep = self._c_weights()
        if fwd_flags is None:
            fwd_flags = _cache.side(self)["c_scheme_hint"][0]
        dev = idx.device
        with torch.cuda.device(dev):
            ws, stream = self._c_workspace(L, cw, B, 4 * H, 4 * W, dev)
            x_hat = torch.empty((B, cw.dims.in_ch, 4 * H, 4 * W), dtype=torch.float32, device=dev)
            _lib.check(L.vqvae_decode_f32(cw, idx.data_ptr(), B, H, W, fwd_flags, x_hat.data_ptr(), ws.data_ptr(), ws.numel(), stream))
        return x_hat
