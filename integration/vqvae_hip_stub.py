"""The binding a maintainer of the reference would add -- standalone: ctypes + torch only, nothing from vqvae_amd/.

    import vqvae_hip_stub as hip                       # next to main.py in the reference tree
    model = VQVAE(128, 32, 2, 512, 64, .25).to('cuda').eval()      # the reference's OWN class (models/vqvae.py:10)
    hip.install(model)                                 # model(x) -> (embedding_loss, x_hat, perplexity) on libvqvae_hip.so
    idx = model.encode(x); x_hat = model.decode_indices(idx, B, 8, 8)     # the index wire format (README.md:56, notebook cells)
    hip.install_quantizer(model.vector_quantization)   # or only VectorQuantizer.forward (models/quantizer.py:29-76)

`install` reads the parameters through `model.state_dict()` -- the 23 keys of SURVEY.md 8b -- so it works on the
reference's classes and on anything that loads its checkpoints.  Inference only (torch.no_grad(), fp32, CUDA tensors),
like the conv kernels behind it.  The C contract is include/vqvae_hip.h; executed by tests/test_integration_stub.py.
"""
import ctypes as C
import os
import types

import torch

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class Dims(C.Structure):                                  # VqvaeDims
    _fields_ = [(n, _i32) for n in ("h_dim", "res_h_dim", "n_res_layers", "n_embeddings", "embedding_dim", "in_ch")] + \
               [("beta", _f32)]


RAW = [("enc0_w", "encoder.conv_stack.0.weight"), ("enc0_b", "encoder.conv_stack.0.bias"),
       ("enc2_w", "encoder.conv_stack.2.weight"), ("enc2_b", "encoder.conv_stack.2.bias"),
       ("enc4_w", "encoder.conv_stack.4.weight"), ("enc4_b", "encoder.conv_stack.4.bias"),
       ("enc_res_w1", "encoder.conv_stack.5.stack.0.res_block.1.weight"),
       ("enc_res_w2", "encoder.conv_stack.5.stack.0.res_block.3.weight"),
       ("pre_w", "pre_quantization_conv.weight"), ("pre_b", "pre_quantization_conv.bias"),
       ("codebook", "vector_quantization.embedding.weight"),
       ("dec0_w", "decoder.inverse_conv_stack.0.weight"), ("dec0_b", "decoder.inverse_conv_stack.0.bias"),
       ("dec_res_w1", "decoder.inverse_conv_stack.1.stack.0.res_block.1.weight"),
       ("dec_res_w2", "decoder.inverse_conv_stack.1.stack.0.res_block.3.weight"),
       ("dec2_w", "decoder.inverse_conv_stack.2.weight"), ("dec2_b", "decoder.inverse_conv_stack.2.bias"),
       ("dec4_w", "decoder.inverse_conv_stack.4.weight"), ("dec4_b", "decoder.inverse_conv_stack.4.bias")]


class RawWeights(C.Structure):                            # VqvaeRawWeights
    _fields_ = [(f, _vp) for f, _ in RAW]


class Weights(C.Structure):                               # VqvaeWeights (19 pointers behind the dims)
    _fields_ = [("dims", Dims)] + [(f"p{i}", _vp) for i in range(19)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.environ.get("VQVAE_HIP_LIB", os.path.join(here, "..", "vqvae_amd", "libvqvae_hip.so"))
        L = C.CDLL(path)                                  # after `import torch`: shares torch's HIP runtime
        L.vqvae_strerror.restype = C.c_char_p
        for name, res, args in (
                ("vqvae_weights_packed_bytes", _sz, [C.POINTER(Dims)]),
                ("vqvae_weights_pack_f32", _i32, [C.POINTER(Dims), C.POINTER(RawWeights), _vp, _sz, C.POINTER(Weights), _vp]),
                ("vqvae_workspace_bytes", _sz, [C.POINTER(Dims), _i64, _i32, _i32]),
                ("vqvae_forward_f32", _i32, [C.POINTER(Weights), _vp, _i64, _i32, _i32, _i32] + [_vp] * 5 + [_sz, _vp, _sz, _vp]),
                ("vqvae_weights_range_check_f32", _i32, [C.POINTER(Dims), C.POINTER(RawWeights), _vp, C.POINTER(_i32), _vp, _sz, _vp]),
                ("vqvae_encode_f32", _i32, [C.POINTER(Weights), _vp, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp, _sz, _vp]),
                ("vqvae_decode_f32", _i32, [C.POINTER(Weights), _vp, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
                ("vqvae_vq_workspace_bytes", _sz, [_i64, _i32, _i32]),
                ("vqvae_vq_forward_f32", _i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32] + [_vp] * 6 + [_sz, _vp])):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    if rc:
        raise RuntimeError(lib().vqvae_strerror(rc).decode())


def pack(model):
    """state_dict -> (Weights, tensors to keep alive).  Call again after the parameters change."""
    sd = model.state_dict()
    keep = {f: sd[k].detach().float().contiguous() for f, k in RAW}
    h = keep["enc4_w"].shape[0]
    K, D = keep["codebook"].shape
    n_res = sum(1 for k in sd if k.startswith("encoder.conv_stack.5.stack.") and k.endswith("res_block.1.weight"))
    dims = Dims(h, keep["enc_res_w1"].shape[0], n_res, K, D, keep["enc0_w"].shape[1], float(model.vector_quantization.beta))
    L = lib()
    nbytes = L.vqvae_weights_packed_bytes(dims)
    if not nbytes:
        raise RuntimeError("model dimensions not supported by libvqvae_hip.so")
    dev = keep["enc0_w"].device
    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    w = Weights()
    raw = RawWeights(**{f: t.data_ptr() for f, t in keep.items()})
    rec = _i32(0)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        _check(L.vqvae_weights_pack_f32(dims, raw, packed.data_ptr(), nbytes, w, st))
        # which product scheme this checkpoint calls for (once per pack; synchronises the stream)
        scratch = torch.empty(16, dtype=torch.float32, device=dev)
        _check(L.vqvae_weights_range_check_f32(dims, raw, None, C.byref(rec), scratch.data_ptr(), 64, st))
    return w, (keep, packed, int(rec.value))


def install(model):
    """Replace `model.forward` (models/vqvae.py:29-44) by one call of vqvae_forward_f32."""
    state = {"w": None}

    def forward(self, x, verbose=False):
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError("libvqvae_hip.so takes CUDA(HIP) fp32 tensors: there is no CPU path")
        if state["w"] is None:
            state["w"] = pack(self)
        w, _keep = state["w"]
        L = lib()
        x = x.contiguous()
        B, _, H, W = x.shape
        with torch.cuda.device(x.device):
            n = L.vqvae_workspace_bytes(w.dims, B, H, W)
            if not n:
                raise RuntimeError(f"image shape {tuple(x.shape)} not supported")
            ws = torch.empty(n, dtype=torch.uint8, device=x.device)
            x_hat = torch.empty_like(x)
            out = torch.empty(2, dtype=torch.float32, device=x.device)
            _check(L.vqvae_forward_f32(w, x.data_ptr(), B, H, W, _keep[2], x_hat.data_ptr(), out.data_ptr(), out.data_ptr() + 4, None,
                                       ws.data_ptr(), n, None, 0, torch.cuda.current_stream(x.device).cuda_stream))
        return out[0], x_hat, out[1]                      # (embedding_loss, x_hat, perplexity), models/vqvae.py:44

    def _ws(w, B, H, W, dev):
        n = lib().vqvae_workspace_bytes(w.dims, B, H, W)
        if not n:
            raise RuntimeError(f"image shape ({B}, ., {H}, {W}) not supported")
        return torch.empty(n, dtype=torch.uint8, device=dev)

    @torch.no_grad()
    def encode(self, x):
        """x -> min_encoding_indices (N, 1) int64: what README.md:56 / the notebook's encode_data keep of a forward.  One call
        (vqvae_encode_f32); on the default shapes no latent map is written at all."""
        if state["w"] is None:
            state["w"] = pack(self)
        w, _keep = state["w"]
        x = x.contiguous()
        B, _, H, W = x.shape
        with torch.cuda.device(x.device):
            ws = _ws(w, B, H, W, x.device)
            idx = torch.empty(B * (H // 4) * (W // 4), 1, dtype=torch.int64, device=x.device)
            _check(lib().vqvae_encode_f32(w, x.data_ptr(), B, H, W, _keep[2], idx.data_ptr(), ws.data_ptr(), ws.numel(), None, 0,
                                          torch.cuda.current_stream(x.device).cuda_stream))
        return idx

    @torch.no_grad()
    def decode_indices(self, idx, B, h, w_):
        """indices -> x_hat: the notebook's generate_samples (one-hot @ embedding.weight -> view -> permute -> decoder,
        visualization.ipynb:358-365) as one call (vqvae_decode_f32).  h, w_: the latent map's size."""
        if state["w"] is None:
            state["w"] = pack(self)
        w, _keep = state["w"]
        idx = idx.contiguous().view(-1)
        K = w.dims.n_embeddings
        if idx.numel() != B * h * w_ or int(idx.min()) < 0 or int(idx.max()) >= K:
            raise IndexError("indices must be B*h*w values in [0, K)")
        with torch.cuda.device(idx.device):
            ws = _ws(w, B, 4 * h, 4 * w_, idx.device)
            x_hat = torch.empty(B, w.dims.in_ch, 4 * h, 4 * w_, dtype=torch.float32, device=idx.device)
            _check(lib().vqvae_decode_f32(w, idx.data_ptr(), B, h, w_, _keep[2], x_hat.data_ptr(), ws.data_ptr(), ws.numel(),
                                          torch.cuda.current_stream(idx.device).cuda_stream))
        return x_hat

    model.forward = types.MethodType(forward, model)
    model.encode = types.MethodType(encode, model)
    model.decode_indices = types.MethodType(decode_indices, model)
    model.hip_repack = lambda: state.update(w=None)       # after load_state_dict / an optimizer step
    return model


def install_quantizer(vq):
    """Replace `VectorQuantizer.forward` (models/quantizer.py:29-76) by vqvae_vq_forward_f32: same 5-tuple."""
    def forward(self, z):
        L = lib()
        z = z.contiguous()
        B, D, H, W = z.shape
        K = self.n_e
        E = self.embedding.weight.detach().contiguous()
        with torch.cuda.device(z.device):
            ws = torch.empty(L.vqvae_vq_workspace_bytes(B * H * W, K, D), dtype=torch.uint8, device=z.device)
            z_q = torch.empty_like(z)
            idx = torch.empty(B * H * W, 1, dtype=torch.int64, device=z.device)
            hist = torch.empty(K, dtype=torch.int32, device=z.device)
            out = torch.empty(2, device=z.device)
            _check(L.vqvae_vq_forward_f32(z.data_ptr(), E.data_ptr(), B, D, H, W, K, self.beta, 0, z_q.data_ptr(), idx.data_ptr(),
                                          hist.data_ptr(), out.data_ptr(), out.data_ptr() + 4, ws.data_ptr(), ws.numel(),
                                          torch.cuda.current_stream(z.device).cuda_stream))
        min_encodings = torch.zeros(B * H * W, K, device=z.device).scatter_(1, idx, 1)          # quantizer.py:55-57
        return out[0], z_q, out[1], min_encodings, idx                                            # quantizer.py:76

    vq.forward = types.MethodType(forward, vq)
    return vq
