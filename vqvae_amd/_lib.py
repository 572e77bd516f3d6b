"""ctypes binding of libvqvae_hip.so (the C ABI declared in include/vqvae_hip.h).

There is deliberately NO fallback here: if the HIP library is missing or cannot be
loaded, importing the product path raises.  torch must be imported first so that
the library binds to the HIP runtime torch has already loaded (one runtime per
process is required for shared streams and device pointers).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads libamdhip64 first)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VQVAE_HIP_LIB_OVERRIDE") or os.path.join(_HERE, "libvqvae_hip.so")   # override: A/B tools only

_i64, _i32, _f32, _vp, _sz = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_size_t

class VqvaeDims(C.Structure):
    _fields_ = [("h_dim", _i32), ("res_h_dim", _i32), ("n_res_layers", _i32), ("n_embeddings", _i32),
                ("embedding_dim", _i32), ("in_ch", _i32), ("beta", _f32)]


_RAW_FIELDS = ["enc0_w", "enc0_b", "enc2_w", "enc2_b", "enc4_w", "enc4_b", "enc_res_w1", "enc_res_w2", "pre_w", "pre_b",
               "codebook", "dec0_w", "dec0_b", "dec_res_w1", "dec_res_w2", "dec2_w", "dec2_b", "dec4_w", "dec4_b"]
_PACKED_FIELDS = ["enc0", "enc0_b", "enc2", "enc2_b", "enc4", "enc4_b", "enc_res_w1", "enc_res_w2", "pre", "pre_b", "codebook",
                  "dec0", "dec0_b", "dec_res_w1", "dec_res_w2", "dec2", "dec2_b", "dec4", "dec4_b"]


class VqvaeRawWeights(C.Structure):
    _fields_ = [(n, _vp) for n in _RAW_FIELDS]


class VqvaeWeights(C.Structure):
    _fields_ = [("dims", VqvaeDims)] + [(n, _vp) for n in _PACKED_FIELDS]


_dimsp, _rawp, _wp = C.POINTER(VqvaeDims), C.POINTER(VqvaeRawWeights), C.POINTER(VqvaeWeights)

# name -> (restype, argtypes); mirrors include/vqvae_hip.h one to one
SIGNATURES = {
    "vqvae_abi_version": (_i32, []),
    "vqvae_source_fingerprint": (C.c_char_p, []),
    "vqvae_strerror": (C.c_char_p, [_i32]),
    "vqvae_profile_enable": (_i32, [_i32]),
    "vqvae_profile_collect": (_i32, [_i32, C.POINTER(C.c_double), C.POINTER(_i32)]),
    "vqvae_calibration_scratch_bytes": (_sz, []),
    "vqvae_calibration_flops": (C.c_double, [_i32]),
    "vqvae_calibration_mfma_f16": (_i32, [_i32, _vp, _sz, _vp]),
    "vqvae_vq_kernel_name": (C.c_char_p, [_i32, _i32, _i32]),
    "vqvae_vq_screen_sweeps": (_i32, [_i32, _i32, _i32]),
    "vqvae_vq_launch_form": (_i32, [_i64, _i32, _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "vqvae_vq_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "vqvae_vq_forward_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vqvae_vq_onehot_f32": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "vqvae_debug_row_sqnorm_f32": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "vqvae_vq_decode_indices_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vqvae_conv_packed_bytes": (_sz, [_i32, _i32, _i32]),
    "vqvae_conv_term_products": (_i32, [_i32] * 6),
    "vqvae_conv_pack_f32": (_i32, [_i32, _vp, _i32, _i32, _vp, _vp]),
    "vqvae_conv_forward_f32": (_i32, [_i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vqvae_res_layer_forward_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vqvae_res_layer_forward_ws_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "vqvae_res_layer_forward_hidden_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vqvae_conv_in_packed_bytes": (_sz, [_i32, _i32]),
    "vqvae_conv_in_pack_f32": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "vqvae_conv_in_forward_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vqvae_conv_taps_packed_bytes": (_sz, [_i32, _i32, _i32]),
    "vqvae_conv_taps_pack_f32": (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "vqvae_conv_taps_forward_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "vqvae_conv_taps_forward_ep_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "vqvae_conv_forward_ep_f32": (_i32, [_i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "vqvae_conv_in_forward_ep_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vqvae_convt_out_packed_bytes": (_sz, [_i32, _i32]),
    "vqvae_convt_out_pack_f32": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "vqvae_convt_out_forward_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vqvae_transpose_f32": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "vqvae_vq_backward_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "vqvae_vq_backward_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32,
                                     _vp, _vp, _vp, _sz, _vp]),
    "vqvae_recon_loss_workspace_bytes": (_sz, []),
    "vqvae_recon_loss_f32": (_i32, [_vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vqvae_recon_loss_backward_f32": (_i32, [_vp, _vp, _i64, _f32, _vp, _vp, _vp]),
    "vqvae_conv_wgrad_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "vqvae_conv_wgrad_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                    _vp, _vp, _sz, _vp]),
    "vqvae_conv_wgrad_ex_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                       _vp, _vp, _sz, _vp]),
    "vqvae_bias_grad_workspace_bytes": (_sz, [_i32]),
    "vqvae_bias_grad_f32": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "vqvae_relu_backward_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "vqvae_weights_packed_bytes": (_sz, [_dimsp]),
    "vqvae_weights_pack_f32": (_i32, [_dimsp, _rawp, _vp, _sz, _wp, _vp]),
    "vqvae_weights_range_check_f32": (_i32, [_dimsp, _rawp, _vp, C.POINTER(C.c_int), _vp, _sz, _vp]),
    "vqvae_workspace_bytes": (_sz, [_dimsp, _i64, _i32, _i32]),
    "vqvae_workspace_ze_offset": (_sz, [_dimsp, _i64, _i32, _i32]),
    "vqvae_resstack_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vqvae_encoder_f32": (_i32, [_wp, _vp, _i64, _i32, _i32, _vp, _vp, _sz, _vp]),
    "vqvae_decoder_f32": (_i32, [_wp, _vp, _i64, _i32, _i32, _vp, _vp, _sz, _vp]),
    "vqvae_encoder_ex_f32": (_i32, [_wp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "vqvae_decoder_ex_f32": (_i32, [_wp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "vqvae_forward_f32": (_i32, [_wp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "vqvae_encode_f32": (_i32, [_wp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp, _sz, _vp]),
    "vqvae_decode_f32": (_i32, [_wp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "vqvae_forward_begin_f32": (_i32, [_wp, _i64, _i32, _i32, _i32, _vp, _sz, _vp, _sz, _vp]),
    "vqvae_forward_part_f32": (_i32, [_wp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "vqvae_forward_end_f32": (_i32, [_wp, _i64, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "vqvae_forward_abort_f32": (_i32, [_vp]),
    "vqvae_gather_rows_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "vqvae_im2col_rows_f32": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "vqvae_gated_activation_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "vqvae_add_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
}

_lib = None


class VqvaeHipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VqvaeHipError(
            f"{LIB_PATH} is missing: build it with `python -m vqvae_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for this path.")
    lib = _open_checked()
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI drifted
        fn.restype, fn.argtypes = res, args
    if lib.vqvae_abi_version() != 9:
        raise VqvaeHipError("libvqvae_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _open_checked():
    """dlopen the in-tree library after making sure it was built from the csrc/ that lies next to it: a library left over from other
    sources (an experiment reverted without a rebuild, a checkout) is rebuilt where hipcc exists and refused where it does not --
    never silently measured or tested.  The library carries its sources' fingerprint as a marked string, read from the FILE: the
    stale one is never loaded.  Not checked: VQVAE_HIP_LIB_OVERRIDE (names an A/B build on purpose), VQVAE_HIP_TRUST_PREBUILT=1
    (a deployment that ships the built library and does not want its sources looked at), and a tree without csrc/*.hip at all
    (an installed package: there is nothing to compare with)."""
    if (os.environ.get("VQVAE_HIP_LIB_OVERRIDE") or os.environ.get("VQVAE_HIP_TRUST_PREBUILT", "0") not in ("", "0")):
        return C.CDLL(LIB_PATH)
    from . import build as _build
    if not _build.have_sources():
        return C.CDLL(LIB_PATH)
    want = _build.source_fingerprint()
    # Several ranks of one node may get here at once.  The comparison, a rebuild and the dlopen all happen under one lock, and
    # the build renames the finished file into place (build.link): nobody maps a half-written library.
    import fcntl
    try:
        lock = open(LIB_PATH + ".lock", "w")
    except OSError:                                   # read-only tree: nobody can be rebuilding it either
        lock = None
    try:
        if lock is not None:
            fcntl.flock(lock, fcntl.LOCK_EX)
        have = _build.library_fingerprint(LIB_PATH)
        if have != want:
            try:
                _build.hipcc()
            except RuntimeError:
                raise VqvaeHipError(f"{LIB_PATH} was built from other sources (fingerprint {have}, csrc/ is {want}) and hipcc is "
                                    "not here to rebuild it: run `python -m vqvae_amd.build -f` where it is, or set "
                                    "VQVAE_HIP_TRUST_PREBUILT=1 to use the library as it is") from None
            _build.build(force=True)
            if _build.library_fingerprint(LIB_PATH) != want:
                raise VqvaeHipError(f"{LIB_PATH}: rebuilt, and its fingerprint still differs from csrc/")
        return C.CDLL(LIB_PATH)
    finally:
        if lock is not None:
            fcntl.flock(lock, fcntl.LOCK_UN)
            lock.close()


ERR_UNSUPPORTED = -3      # VQVAE_ERR_UNSUPPORTED (include/vqvae_hip.h)


def check(code: int):
    if code != 0:
        raise VqvaeHipError(f"libvqvae_hip: {load().vqvae_strerror(code).decode()} (code {code})")


def vq_kernel_name(K: int, D: int, flags: int = 0x1) -> str:
    """kernel vqvae_vq_forward_f32 launches for this shape (default flags: row-major rows, the fused path's layout)"""
    return load().vqvae_vq_kernel_name(K, D, flags).decode()


def vq_sweeps(K: int, D: int, flags: int = 0x1) -> int:
    return load().vqvae_vq_screen_sweeps(K, D, flags)


def vq_launch_form(n_rows: int, K: int, D: int, HW: int = 64, flags: int = 0x1):
    """(waves per CU, rows per unit, pooled tail units in per cent) of vq_track_kernel_d64 for this problem on the current device,
    or None where another kernel runs"""
    w, r, p = _i32(), _i32(), _i32()
    if load().vqvae_vq_launch_form(n_rows, K, D, HW, flags, C.byref(w), C.byref(r), C.byref(p)) != 0:
        return None
    return w.value, r.value, p.value


def vq_kernel_instance(n_rows: int, K: int, D: int, HW: int = 64, flags: int = 0x1) -> str:
    """the kernel's name with the template instance that runs for n_rows rows, e.g. vq_track_kernel_d64<16, false, 1>"""
    name = vq_kernel_name(K, D, flags)
    f = vq_launch_form(n_rows, K, D, HW, flags)
    if f is None:
        return name
    # (codebooks of exactly sixteen 32-code tiles, K in 481 .. 512, run instances with the sweep fully unrolled -- every launch form
    # but 64-row units on eight waves of row-major rows, which the rule no longer picks for such codebooks)
    unrolled = ", 16" if 480 < K <= 512 and not (f[0] == 8 and f[1] == 64 and flags & 0x1) and f[0] != 12 else ""
    return f"{name}<{f[0]}, {'false' if flags & 0x1 else 'true'}, {f[1] // 32}{unrolled}>" + (f" (last {f[2]} % of the units pooled)" if f[2] else "")


PROF_IDS = {"vq_main": 0, "conv_igemm": 1, "res_layer": 2, "conv_in": 3, "conv_out": 4}


def profile_enable(on: bool):
    check(load().vqvae_profile_enable(1 if on else 0))


def profile_collect(name: str):
    """-> (total_ms, launches) of one instrumented kernel since the last collect (synchronises)."""
    ms, n = C.c_double(), _i32()
    check(load().vqvae_profile_collect(PROF_IDS[name], C.byref(ms), C.byref(n)))
    return ms.value, n.value
