"""CPU tests: the C-ABI library builds, loads, and exports exactly what include/vqvae_hip.h declares.
No compute calls here (no GPU in this container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from vqvae_amd import build
    path = build.build()
    import torch  # noqa: F401  HIP runtime first
    return ctypes.CDLL(path)


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include/vqvae_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(vqvae_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 6
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vqvae_hip.h but not exported"


def test_binding_table_matches_header():
    from vqvae_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_no_torch_types_in_abi():
    hdr = open(os.path.join(ROOT, "include/vqvae_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)          # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code


def test_argument_errors_without_gpu(lib):
    """Argument validation happens before any HIP call."""
    lib.vqvae_strerror.restype = ctypes.c_char_p
    lib.vqvae_vq_workspace_bytes.restype = ctypes.c_size_t
    lib.vqvae_vq_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    assert lib.vqvae_abi_version() == 9
    assert lib.vqvae_vq_workspace_bytes(2048, 512, 64) > 512 * 64 * 4
    assert lib.vqvae_vq_workspace_bytes(2048, 512, 48) > 0           # any width up to 256 (round 5: the exact-fp32 vector kernel)
    assert lib.vqvae_vq_workspace_bytes(2048, 512, 257) == 0         # unsupported D
    assert b"NULL" in lib.vqvae_strerror(-1)
    f = lib.vqvae_vq_forward_f32
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int] + \
        [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p]
    assert f(None, None, 1, 64, 8, 8, 512, 0.25, 0, None, None, None, None, None, None, 0, None) == -1
    one = ctypes.c_void_p(16)
    assert f(one, one, 0, 64, 8, 8, 512, 0.25, 0, one, one, one, one, one, one, 1 << 30, None) == -2
    assert f(one, one, 1, 257, 8, 8, 512, 0.25, 0, one, one, one, one, one, one, 1 << 30, None) == -3    # (D <= 256: any width, round 5)
    assert f(one, one, 1, 64, 8, 8, 512, 0.25, 0, one, one, one, one, one, one, 16, None) == -4


def test_conv_and_training_argument_errors_without_gpu():
    """Same for the conv / residual / training entry points: NULL, shape, unsupported and alignment errors
    come back as negative codes before anything is launched."""
    from vqvae_amd import _lib
    L = _lib.load()
    a = 256                                   # a fake, 16-byte aligned "device pointer" (never dereferenced)
    assert L.vqvae_conv_packed_bytes(1, 128, 128) > 0
    assert L.vqvae_conv_packed_bytes(9, 128, 128) == 0                                   # unknown kind
    assert L.vqvae_conv_packed_bytes(0, 64, 128) > 2 * L.vqvae_conv_packed_bytes(1, 64, 128) * 16 // 9 // 2
    assert L.vqvae_conv_forward_f32(1, None, a, a, 1, 8, 8, 128, 128, 0, a, None) == -1
    assert L.vqvae_conv_forward_f32(1, a, a, a, 0, 8, 8, 128, 128, 0, a, None) == -2
    assert L.vqvae_conv_forward_f32(1, a, a, a, 1, 8, 8, 126, 128, 0, a, None) == -3        # Cin % 4
    assert L.vqvae_conv_forward_f32(1, a + 4, a, a, 1, 8, 8, 128, 128, 0, a, None) == -3    # misaligned x
    assert L.vqvae_conv_forward_f32(0, a, a, a, 1, 7, 8, 128, 128, 0, a, None) == -3        # odd H for stride 2
    assert L.vqvae_res_layer_forward_f32(a, a, a, 1, 8, 8, 96, 32, 0, a + 4096, None) == -3   # C not in {32,64,128}
    assert L.vqvae_res_layer_forward_f32(a, a, a, 1, 8, 8, 128, 64, 0, a + 4096, None) == -3  # res_h > 32
    assert L.vqvae_res_layer_forward_f32(a, a, a, 1, 8, 8, 128, 32, 0, a, None) == -3         # in place
    assert L.vqvae_res_layer_forward_hidden_f32(a, a, a, 1, 8, 8, 128, 32, 0, a + 4096, None, None) == -1     # no hidden buffer
    assert L.vqvae_res_layer_forward_hidden_f32(a, a, a, 1, 16, 16, 128, 32, 0, a + 4096, a + 8192, None) == -3   # only 8x8 maps
    assert L.vqvae_res_layer_forward_hidden_f32(a, a, a, 1, 8, 8, 128, 16, 0, a + 4096, a + 8192, None) == -3     # only res_h = 32
    assert L.vqvae_conv_in_forward_f32(a, a, a, 1, 31, 32, 3, 64, 0, a, None) == -3
    assert L.vqvae_conv_in_forward_f32(a, a, a, 1, 32, 32, 2, 64, 0, a, None) == -3           # Cin not in {1,3,4}
    assert L.vqvae_convt_out_forward_f32(a, a, a, 1, 16, 16, 64, 5, 0, a, None) == -3            # Cout > 4
    assert L.vqvae_vq_backward_workspace_bytes(2048, 512, 64) > 2048 * 16
    assert L.vqvae_vq_backward_workspace_bytes(2048, 20000, 64) == 0
    assert L.vqvae_vq_backward_f32(a, a, None, None, None, 1, 64, 8, 8, 512, 0.25, 0, a, a, a, 1 << 30, None) == -1
    assert L.vqvae_vq_backward_f32(a, a, a, None, None, 1, 64, 8, 8, 512, 0.25, 0, None, a, a, 16, None) == -4
    assert L.vqvae_recon_loss_f32(a, a, 0, 1.0, None, None, a, a, 1 << 20, None) == -2
    assert L.vqvae_recon_loss_f32(a, a + 4, 16, 1.0, None, None, a, a, 1 << 20, None) == -3
    assert L.vqvae_recon_loss_f32(a, a, 16, 1.0, None, None, a, a, 8, None) == -4
    assert L.vqvae_transpose_f32(None, 1, 8, 8, a, None) == -1
    assert L.vqvae_conv_wgrad_workspace_bytes(128, 128, 3) == 128 * 9 * 128 * 128 * 4      # 512 workgroups over four 64 x 64 tiles
    assert L.vqvae_conv_wgrad_workspace_bytes(100, 60, 3) == 64 * 9 * 100 * 60 * 4           # generic kernel: 64 pixel ranges
    assert L.vqvae_conv_wgrad_workspace_bytes(128, 128, 5) == 0
    assert L.vqvae_conv_wgrad_f32(a, None, 1, 8, 8, 128, 8, 8, 128, 3, 1, 1, 0, a, a, 1 << 30, None) == -1
    assert L.vqvae_conv_wgrad_f32(a, a, 1, 8, 8, 128, 8, 8, 128, 5, 1, 1, 0, a, a, 1 << 30, None) == -3
    assert L.vqvae_conv_wgrad_f32(a, a, 1, 8, 8, 128, 8, 8, 128, 3, 1, 1, 0, a, a, 16, None) == -4
    assert L.vqvae_conv_wgrad_f32(a + 4, a, 1, 8, 8, 128, 8, 8, 128, 3, 1, 1, 0, a, a, 1 << 30, None) == -3
    assert L.vqvae_bias_grad_f32(a, 1, 64, 300, 0, a, a, 1 << 30, None) == -3
    assert L.vqvae_bias_grad_f32(a, 0, 64, 128, 0, a, a, 1 << 30, None) == -2
    assert L.vqvae_relu_backward_f32(a, None, 16, a, None) == -1
    assert L.vqvae_relu_backward_f32(a, a + 4, 16, a, None) == -3


def test_product_path_does_not_import_oracle():
    """The shipped package must not import, link or execute oracle/ (test infrastructure)."""
    pat = re.compile(r"(^|\s)(from|import)\s+oracle\b|libvqvae_oracle|oracle[/.](c_oracle|torch_port|vqvae_oracle)")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vqvae_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not pat.search(src), f"{fn} reaches into oracle/"


def test_host_side_plans_without_gpu():
    """Pure host logic of round 2: which quantizer kernel / conv product scheme a shape gets, and how the packed-weight and
    workspace sizes are laid out (no HIP calls)."""
    from vqvae_amd import _lib
    L = _lib.load()
    # quantizer dispatch (row-major flag 0x1): resident-image sweep for small codebooks, streamed image beyond, exact on request
    assert _lib.vq_kernel_name(512, 64) == "vq_track_kernel_d64"
    assert _lib.vq_kernel_name(512, 64, 0x1 | 0x10) == "unsupported"              # round 2's tracker (flags 0x10 / 0x20): removed in round 4
    assert _lib.vq_kernel_name(1024, 64) == "vq_track_kernel_d64"                 # 128 KiB image beside four waves' tiles (round 4)
    assert _lib.vq_kernel_name(1056, 64) == "vq_stream_sweep_kernel"
    assert _lib.vq_kernel_name(8192, 128) == "vq_stream_sweep_kernel"
    assert _lib.vq_kernel_name(512, 64, 0x1 | 0x8) == "vq_filter_kernel_d64"
    assert _lib.vq_kernel_name(512, 64, 0x0) == "vq_track_kernel_d64"           # NCHW rows (maps of 64 k pixels; round 4)
    assert _lib.vq_kernel_name(1024, 64, 0x0) == "vq_track_kernel_d64"          # NCHW, K up to 1024: four waves, half-tile transposition (round 5)
    assert _lib.vq_kernel_name(1056, 64, 0x0) == "vq_exact_kernel"              # NCHW, codebook beyond the LDS-resident image
    assert _lib.vq_kernel_name(1024, 64, 0x8) == "vq_filter_kernel_d64"         # (round 1's two-sweep kernel stays reachable by its flag)
    assert _lib.vq_kernel_name(512, 256) == "vq_exact_kernel"
    # the stream-tracker kernel's launch forms (256 CUs assumed where no device is present): config 2 / config 3 / many rows / K = 1024 /
    # the module's NCHW layout; the rule scales with the CU count, so only sizes far from its edges are pinned here
    cus = 256
    assert _lib.vq_launch_form(8 * cus * 32, 512, 64) == (8, 32, 0)               # BASELINE config 2 on 256 CUs: 32-row units, eight waves
    assert _lib.vq_launch_form(32 * cus * 32, 512, 64) == (16, 32, 0)             # config 3: sixteen waves
    assert _lib.vq_launch_form(256 * cus * 32, 512, 64) == (16, 32, 25)           # many rows: sixteen waves as well (round 5), pooled tail
    assert _lib.vq_launch_form(256 * cus * 32, 600, 64) == (8, 64, 25)            # a codebook whose image leaves no room for sixteen tiles: 64-row units
    assert _lib.vq_launch_form(32 * cus * 32, 1024, 64) == (4, 32, 25)            # config 4's codebook: four waves (eight units per wave here: pooled tail)
    assert _lib.vq_launch_form(256 * cus * 32, 1024, 64)[:2] == (4, 32)
    assert _lib.vq_launch_form(8 * cus * 32, 512, 64, 64, 0x0) == (8, 32, 0)      # NCHW 8x8 maps, few rows
    assert _lib.vq_launch_form(256 * cus * 32, 512, 64, 64, 0x0)[:2] == (8, 64)
    assert _lib.vq_launch_form(256 * cus * 32, 512, 64, 96, 0x0)[:2] == (8, 32)   # 32 (2 k + 1) pixels: 32-position units at any size
    assert _lib.vq_launch_form(4096, 512, 64, 49, 0x0) is None                    # 7x7 maps: another kernel
    assert _lib.vq_launch_form(4096, 1024, 64, 64, 0x0)[:2] == (4, 32) and _lib.vq_launch_form(4096, 2048, 64) is None
    assert _lib.vq_launch_form(32 * cus * 32, 512, 64, 64, 0x1 | 0x100) == (8, 64, 0)     # forced forms
    assert _lib.vq_kernel_instance(32 * cus * 32, 512, 64) == "vq_track_kernel_d64<16, false, 1, 16>"      # (the unrolled-sweep instance, round 5)
    assert _lib.vq_kernel_instance(32 * cus * 32, 448, 64) == "vq_track_kernel_d64<16, false, 1>"
    assert _lib.vq_sweeps(512, 64) == 1 and _lib.vq_sweeps(512, 64, 0x1 | 0x8) == 2 and _lib.vq_sweeps(512, 256) == 0
    # the streamed kernels' scratch does not grow with the row count (slabs of 2^18 rows)
    a, b = L.vqvae_vq_workspace_bytes(1000, 8192, 128), L.vqvae_vq_workspace_bytes(10 ** 8, 8192, 128)
    assert a == b and a > (1 << 18) * 128 * 2
    assert L.vqvae_vq_workspace_bytes(1000, 512, 64) < 4 << 20                   # resident-image kernel: no row scratch
    # conv product schemes
    assert L.vqvae_conv_term_products(1, 8, 8, 128, 128, 0) == 3
    assert L.vqvae_conv_term_products(1, 56, 56, 128, 128, 0) == 6
    assert L.vqvae_conv_term_products(1, 8, 8, 128, 128, 8) == 6 and L.vqvae_conv_term_products(1, 8, 8, 128, 128, 4) == 1
    assert L.vqvae_conv_term_products(7, 8, 8, 128, 128, 0) == 0                  # unknown kind
    # packed weights: fp32 image + three-term bf16 image + header + two-term fp16 image (4x4 s2: both in two chunk orders)
    cells = 9 * 4 * 4                                                            # taps x Cin chunks x Cout tiles
    # header (round 4): 64 ints + one float (2^-kw[c]) and one int (kw[c]) per output channel of the 32-channel tiles
    hdr = lambda ntile: 256 + 256 * ntile
    assert L.vqvae_conv_packed_bytes(1, 128, 128) == cells * (4096 + 6144) + hdr(4) + cells * 4096
    cells = 16 * 2 * 4
    assert L.vqvae_conv_packed_bytes(0, 64, 128) == cells * (4096 + 2 * 6144) + hdr(4) + 2 * cells * 4096
    # [fp32][three-term bf16][header][two-term fp16][A-operand image of the fused decoder tail: 16 KiB]
    assert L.vqvae_convt_out_packed_bytes(64, 3) == 2 * 2 * (4096 + 6144) + hdr(1) + 2 * 2 * 4096 + 16384
    # whole-path workspace covers two activation buffers, the latents and the two maxima regions
    dims = _lib.VqvaeDims(128, 32, 2, 512, 64, 3, 0.25)
    ws = L.vqvae_workspace_bytes(dims, 4096, 32, 32)
    act = 4096 * 16 * 16 * 64 * 4
    assert ws > 2 * act + 2 * 4096 * 64 * 64 * 4 + 2 * (4 + 2) * 4096 * 4
    assert L.vqvae_workspace_bytes(dims, 4096, 30, 32) == 0


def test_modules_pickle_and_deepcopy_like_the_reference(tmp_path):
    """The reference's VQVAE is a plain nn.Module: torch.save(model) and copy.deepcopy(model) (EMA copies,
    checkpoint-by-module) work.  Runtime caches live in a weak side table (vqvae_amd/_cache.py), never in __dict__."""
    import copy
    import io
    import torch
    from vqvae_amd.modules import VQVAE
    torch.manual_seed(0)
    m = VQVAE(128, 32, 2, 512, 64, 0.25)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    m3 = copy.deepcopy(m)
    for a in (m2, m3):
        for (k1, v1), (k2, v2) in zip(m.state_dict().items(), a.state_dict().items()):
            assert k1 == k2 and torch.equal(v1, v2)
        # the aliased residual weights stay aliased in the copy (models/residual.py:44-45)
        st = a.encoder.conv_stack[5].stack
        assert st[0].res_block[1].weight is st[1].res_block[1].weight
    assert not any(k.startswith("_c_") or "vqvae_amd" in k for k in m.__dict__)


def test_library_carries_the_fingerprint_of_its_sources_and_a_stale_one_is_refused(monkeypatch):
    """Round 5: a library left over from OTHER sources (an experiment reverted without a rebuild) must never be the thing tests and
    bench measure.  The build links the sources' fingerprint in; `_lib.load()` reads it from the file and rebuilds -- or, without
    hipcc, refuses."""
    from vqvae_amd import _lib, build
    want = build.source_fingerprint()
    assert build.library_fingerprint() == want and not build.stale()
    assert _lib.load().vqvae_source_fingerprint().decode() == want
    # the same check, failing: other sources, no compiler
    monkeypatch.setattr(build, "source_fingerprint", lambda: "0" * 16)
    monkeypatch.setattr(build, "hipcc", lambda: (_ for _ in ()).throw(RuntimeError("no hipcc")))
    monkeypatch.delenv("VQVAE_HIP_LIB_OVERRIDE", raising=False)
    with pytest.raises(_lib.VqvaeHipError, match="built from other sources"):
        _lib._open_checked()


def test_fingerprint_ignores_files_that_are_not_sources_and_prebuilt_trees_can_opt_out(monkeypatch, tmp_path):
    """only csrc/*.hip and *.h make the fingerprint (an editor backup or a patch's .orig file must not make the library
    'stale'); VQVAE_HIP_TRUST_PREBUILT=1 and a tree without csrc/ skip the comparison; the link goes through a temporary name."""
    import os
    from vqvae_amd import _lib, build
    want = build.source_fingerprint()
    junk = os.path.join(build.CSRC, "vq_track.hip.orig~")
    try:
        with open(junk, "w") as f:
            f.write("not a source")
        assert build.source_fingerprint() == want
    finally:
        os.remove(junk)
    # other sources + no compiler, but the deployment says "use it as it is"
    monkeypatch.setattr(build, "source_fingerprint", lambda: "0" * 16)
    monkeypatch.setattr(build, "hipcc", lambda: (_ for _ in ()).throw(RuntimeError("no hipcc")))
    monkeypatch.delenv("VQVAE_HIP_LIB_OVERRIDE", raising=False)
    monkeypatch.setenv("VQVAE_HIP_TRUST_PREBUILT", "1")
    assert _lib._open_checked() is not None
    monkeypatch.delenv("VQVAE_HIP_TRUST_PREBUILT")
    monkeypatch.setattr(build, "have_sources", lambda: False)            # an installed tree: nothing to compare with
    assert _lib._open_checked() is not None
    # build.link never leaves a partial file under the final name
    import inspect
    assert "os.replace" in inspect.getsource(build.link)


def test_variant_builds_export_every_symbol_load_binds():
    """tools/build_variant.py / build_src_variant.sh link the generated fingerprint object too, so a library named by
    VQVAE_HIP_LIB_OVERRIDE passes `_lib.load()`'s symbol loop (the A/B tooling of profiles/ depends on it)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from vqvae_amd import build
    build.build()
    out = subprocess.check_output(["bash", os.path.join(root, "tools/build_src_variant.sh"), "capi.hip", "capitest"], cwd=root).decode()
    lib = os.path.join(root, out.strip().splitlines()[-1])
    code = ("import os, sys; sys.path.insert(0, %r); from vqvae_amd import _lib; L = _lib.load(); "
            "assert _lib.LIB_PATH.endswith('libvqvae_capitest.so'); print(L.vqvae_source_fingerprint().decode())" % root)
    got = subprocess.check_output([sys.executable, "-c", code], env={**os.environ, "VQVAE_HIP_LIB_OVERRIDE": lib}).decode().strip()
    assert got == build.source_fingerprint()
