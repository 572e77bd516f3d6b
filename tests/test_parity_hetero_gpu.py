"""GPU parity tests (-m gpu) on UNFRIENDLY data (VERDICT r3 "next round" item 1): trained-checkpoint-like weights whose
channels differ by up to six decades (tests/hetero.py) and images with in-image outliers, through the whole-path C entry
points vqvae_encoder_f32 / vqvae_decoder_f32 / vqvae_forward_f32 at BASELINE config-3 dimensions, against the
reference's algorithm (oracle/torch_port.py, bitwise the imported reference).

Tolerances, PER OUTPUT CHANNEL (a tolerance relative to the tensor's or the image's maximum would hide a relative loss in a
small channel):
  * z_e and x_hat vs the fp32 reference:  |got - ref| <= 1e-5 * max|ref channel| + 1e-4 * |ref|   (channel maximum over the batch);
  * the same against the fp64 evaluation of the same network, per (image, channel), reported next to the fp32 reference's
    own distance from fp64 (1e-6 ... 6e-6 of the (image, channel) maximum on these weights: the reference is no closer to the
    truth than that) and bounded by 2e-5 -- with ONE stated exception: trained-like weights AND in-image outliers TOGETHER
    under the two-term fp16 scheme.  An activation there can sit 10^4 (outlier) x 10^6 (channel spread) below its image's
    maximum, beyond what one power-of-two scale per image leaves of fp16's 2^-17 full-precision window (the
    per-channel error bound): measured 1.9e-4 of the (image, channel) maximum, bounded here by 1e-3 and PRINTED; the per-channel
    tolerance of the first bullet still holds, and the three-term bf16 / exact-fp32 schemes (VQVAE_FWD_CONV_BF16_SPLIT /
    _EXACT_FP32: the escape hatch for such data) stay at 4e-6 on the same case;
  * indices: bit-exact against the oracle's quantizer run on the DEVICE's z_e bits; flips against the reference's indices
    are counted, printed, and each must be explained by the z_e tolerance (fp64 gap <= 8 eps32 (|z|^2 + |e|^2)
    + 2 sum_c tol_c |e_a,c - e_b,c|).
The product scheme under test is whatever the whole path selects (two-term fp16 with per-output-channel weight scales and
per-image activation scales by default; `scheme` parametrises the whole-path flags for the three-term bf16 and the
exact-fp32 MFMA kernels).
"""
import numpy as np
import pytest
import torch

from tests import hetero

pytestmark = pytest.mark.gpu

DIMS = (128, 32, 2, 512, 64)          # h_dim, res_h, n_res, K, D: main.py defaults = BASELINE config 3
B = 64


def dev():
    return torch.device("cuda:0")


def _state(weights):
    from oracle import torch_port
    sd0 = torch_port.init_state_dict(DIMS[0], DIMS[1], DIMS[3], DIMS[4], seed=0, n_res_layers=DIMS[2])
    kind, seed = weights
    if kind == "default":
        return {k: v.clone() for k, v in sd0.items()}
    if kind == "coupled":
        return hetero.rescale_coupled(sd0, seed, 3.0, DIMS[2])
    if kind == "independent":
        return hetero.rescale_independent(sd0, seed, 3.0, DIMS[2])
    raise KeyError(kind)


def _images(kind):
    if kind == "normal":
        return torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(77))
    return hetero.outlier_images(B, 78, kind)


def _scheme_flags(scheme):
    from vqvae_amd import functional as F
    return {"fp16x2": 0, "bf16x3": F.FWD_CONV_BF16_SPLIT, "fp32": F.FWD_CONV_EXACT_FP32}[scheme]


def _model(sd):
    from vqvae_amd import conv
    from vqvae_amd.modules import VQVAE
    conv.set_conv_backend("hip")
    m = VQVAE(DIMS[0], DIMS[1], DIMS[2], DIMS[3], DIMS[4], 0.25).eval()
    m.load_state_dict(sd)
    return m.to(dev())


def _reference(sd, x):
    from oracle import torch_port
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        z_e = torch_port.encode(sd, x.clone(), DIMS[2])
        z_e64 = torch_port.encode(sd64, x.double(), DIMS[2])
        loss, z_q, ppl, _, idx = torch_port.quantize(z_e, sd["vector_quantization.embedding.weight"], 0.25)
        x_hat = torch_port.decode(sd, z_q.clone(), DIMS[2])
        x_hat64 = torch_port.decode(sd64, z_q.double(), DIMS[2])
    return dict(z_e=z_e, z_e64=z_e64, z_q=z_q, idx=idx, x_hat=x_hat, x_hat64=x_hat64, loss=loss, ppl=ppl)


def _vs_fp64(got, ref32, ref64, what, lim=2e-5):
    got, ref32, ref64 = (np.asarray(a, np.float64) for a in (got, ref32, ref64))
    cmax = np.maximum(np.abs(ref64).max(axis=(2, 3), keepdims=True), 1e-300)
    ours = float((np.abs(got - ref64) / cmax).max())
    theirs = float((np.abs(ref32 - ref64) / cmax).max())
    assert ours <= lim, f"{what}: {ours:.3g} of the (image, channel) maximum away from the fp64 network (fp32 reference: {theirs:.3g})"
    return ours, theirs


CASES = [
    (("default", 0), "pixel"), (("default", 0), "channel"), (("default", 0), "mixed"),
    (("coupled", 1), "normal"), (("coupled", 2), "mixed"),
    (("independent", 1), "normal"), (("independent", 2), "normal"), (("independent", 3), "mixed"),
]


@pytest.mark.parametrize("scheme", ["fp16x2", "bf16x3", "fp32"])
@pytest.mark.parametrize("weights,images", CASES, ids=[f"{w[0]}{w[1]}-{i}" for w, i in CASES])
def test_whole_path_on_heterogeneous_channel_scales(weights, images, scheme, capsys):
    from oracle import torch_port
    from vqvae_amd import _lib
    L = _lib.load()
    flags = _scheme_flags(scheme)
    sd = _state(weights)
    x = _images(images)
    ref = _reference(sd, x)
    m = _model(sd)
    xd = x.to(dev()).contiguous()
    cw, _keep = m._c_weights()
    nws = L.vqvae_workspace_bytes(cw.dims, B, 32, 32)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev())
    st = torch.cuda.current_stream().cuda_stream
    D = DIMS[4]
    with torch.no_grad():
        z_e = torch.empty(B, 8, 8, D, device=dev())
        _lib.check(L.vqvae_encoder_ex_f32(cw, xd.data_ptr(), B, 32, 32, flags, z_e.data_ptr(), ws.data_ptr(), nws, st))
        zq_rows = ref["z_q"].to(dev()).permute(0, 2, 3, 1).contiguous()
        x_hat_dec = torch.empty_like(xd)
        _lib.check(L.vqvae_decoder_ex_f32(cw, zq_rows.data_ptr(), B, 8, 8, flags, x_hat_dec.data_ptr(), ws.data_ptr(), nws, st))
        loss, x_hat, ppl, idx = m._forward_c(xd, want_idx=True, fwd_flags=flags)
    torch.cuda.synchronize()

    # --- encoder: z_e per output channel
    ze = z_e.permute(0, 3, 1, 2).cpu().numpy()
    w_ze = hetero.per_channel_check(ze, ref["z_e"].numpy(), f"z_e [{scheme}]", per_image=False)
    both = weights[0] != "default" and images != "normal" and scheme == "fp16x2"       # the stated exception (module docstring)
    o_ze, t_ze = _vs_fp64(ze, ref["z_e"].numpy(), ref["z_e64"].numpy(), f"z_e [{scheme}]", lim=1e-3 if both else 2e-5)
    # --- decoder on the reference's z_q bits: x_hat per output channel
    w_xh = hetero.per_channel_check(x_hat_dec.cpu().numpy(), ref["x_hat"].numpy(), f"x_hat(decoder) [{scheme}]", per_image=False)
    o_xh, t_xh = _vs_fp64(x_hat_dec.cpu().numpy(), ref["x_hat"].numpy(), ref["x_hat64"].numpy(), f"x_hat(decoder) [{scheme}]",
                          lim=1e-3 if both else 2e-5)

    # --- indices: exact on the device's own z_e bits against the C oracle (the rounding-order specification of SURVEY.md A.1,
    # pinned to the reference in the build container and the same code on every host).  torch's own CPU kernels on THIS host
    # are compared too and the disagreement with the specification printed: ATen picks its vector width by the host's ISA
    # ("properties of this torch build on this host ISA"), and with channels six decades apart the distances
    # of many codes tie in fp32, so a different sum-of-squares order moves an argmin
    from oracle import c_oracle
    cbk = sd["vector_quantization.embedding.weight"]
    own = c_oracle.vq_forward(np.ascontiguousarray(ze), cbk.numpy(), 0.25)["idx"].reshape(-1)
    with torch.no_grad():
        own_t = torch_port.quantize(torch.from_numpy(ze).contiguous(), cbk, 0.25)[4].numpy().reshape(-1)
    host_disagree = int((own != own_t).sum())
    got = idx.cpu().numpy().reshape(-1)
    if not np.array_equal(got, own):
        rows = np.nonzero(got != own)[0]
        zrows = np.transpose(ze, (0, 2, 3, 1)).reshape(-1, D)
        diag = []
        for r in rows[:4]:
            one = c_oracle.vq_forward(np.ascontiguousarray(zrows[r].reshape(1, D, 1, 1)), cbk.numpy(), 0.25, want_dist=True)
            dd = one["dist"][0]
            diag.append(f"row {r} (image {r // 64}): device {got[r]} oracle {own[r]} fp32 d[device]={dd[got[r]]!r} d[oracle]={dd[own[r]]!r} "
                        f"min={dd.min()!r} argmin={int(dd.argmin())} ties_at_min={int((dd == dd.min()).sum())} |z|^2={float((zrows[r].astype(np.float64) ** 2).sum()):.6g} "
                        f"finite={bool(np.isfinite(zrows[r]).all())} max|z|={float(np.abs(zrows[r]).max()):.4g}")
        raise AssertionError(f"{len(rows)} indices differ from the C oracle on the device's own z_e bits (torch on this host differs from the "
                             f"C oracle on {host_disagree} rows): " + " ;; ".join(diag))
    # ... and every flip against the reference's indices explained by the z_e tolerance
    want = ref["idx"].numpy().reshape(-1)
    flips = np.nonzero(got != want)[0]
    zr = ref["z_e64"].permute(0, 2, 3, 1).reshape(-1, D).numpy()
    e = cbk.double().numpy()
    cmax = np.abs(ref["z_e"].numpy()).max(axis=(0, 2, 3))                    # per channel over the batch
    worst_flip = 0.0
    for r in flips:
        d = ((zr[r][None, :] - e) ** 2).sum(1)
        tol_c = 1e-5 * cmax + 1e-4 * np.abs(zr[r])
        bound = 8 * 2.0 ** -24 * ((zr[r] ** 2).sum() + (e[want[r]] ** 2).sum()) + 2 * (tol_c * np.abs(e[got[r]] - e[want[r]])).sum()
        worst_flip = max(worst_flip, abs(d[got[r]] - d[want[r]]) / bound)
    assert worst_flip <= 1.0, f"an index flip is not explained by the z_e tolerance: gap = {worst_flip:.3g} x the bound"
    # (the stated fp16 exception -- trained-like weights and in-image outliers together -- moves more near-ties: 18 of 4096 measured)
    assert len(flips) <= max(2, (1e-2 if both else 2e-3) * got.size), f"{len(flips)} index flips in {got.size} rows"

    # --- the whole forward: x_hat on images without a flip, per channel; loss / perplexity when nothing flipped
    clean = np.setdiff1d(np.arange(B), np.unique(flips // 64))
    w_e2e = hetero.per_channel_check(x_hat.cpu().numpy()[clean], ref["x_hat"].numpy()[clean], f"x_hat(forward) [{scheme}]", per_image=False)
    if len(flips) == 0:
        np.testing.assert_allclose(loss.item(), float(ref["loss"]), rtol=2e-5)
        np.testing.assert_allclose(ppl.item(), float(ref["ppl"]), rtol=1e-5)
    with capsys.disabled():
        print(f"\n   [{weights[0]}{weights[1]} / {images} / {scheme}] worst error in units of the channel maximum: z_e {w_ze:.2e}, "
              f"x_hat(dec) {w_xh:.2e}, x_hat(fwd) {w_e2e:.2e};  vs fp64 per (image, channel): z_e {o_ze:.2e} (fp32 reference "
              f"{t_ze:.2e}), x_hat {o_xh:.2e} ({t_xh:.2e});  {len(flips)} index flips / {got.size} rows; torch on this host vs the C oracle on the same z_e bits: {host_disagree} rows differ")


@pytest.mark.parametrize("weights,images", CASES, ids=[f"{w[0]}{w[1]}-{i}" for w, i in CASES])
def test_range_guard_routes_unfriendly_checkpoints_to_the_three_term_scheme(weights, images, capsys):
    """Round 5 (VERDICT r4 "missing" 6): the caller no longer has to KNOW that a checkpoint defeats the two-term fp16 products.
    vqvae_weights_range_check_f32 measures the spread of every layer's input channels once per weight version; the module's
    default call (no scheme named) follows its recommendation.  Default-initialised weights: the guard stays quiet and the default
    call IS the two-term path, bit for bit.  Trained-like weights (tests/hetero.py, channels six decades apart): the guard fires
    and the default call IS the three-term bf16 path, bit for bit -- whose distance from fp64 the test above bounds by 2e-5 on
    exactly these weights and images, outliers included.  The stated exception of that test is therefore only reachable by NAMING
    the two-term scheme on such a checkpoint."""
    from vqvae_amd import functional as F
    m = _model(_state(weights))
    xd = _images(images).to(dev()).contiguous()
    flags, spreads = m.scheme_hint()
    trained_like = weights[0] != "default"
    assert flags == (F.FWD_CONV_BF16_SPLIT if trained_like else 0), (flags, spreads)
    assert (max(spreads[1:]) > 10.0) == trained_like, spreads
    with torch.no_grad():
        auto = m._forward_c(xd, want_idx=True)
        named = m._forward_c(xd, want_idx=True, fwd_flags=F.FWD_CONV_BF16_SPLIT if trained_like else 0)
        idx_auto = m.encode(xd)
    torch.cuda.synchronize()
    assert torch.equal(auto[3], named[3]) and torch.equal(auto[1].view(torch.int32), named[1].view(torch.int32))
    assert auto[0].item() == named[0].item() and auto[2].item() == named[2].item()
    assert torch.equal(idx_auto, auto[3])                      # encode follows the same recommendation
    with capsys.disabled():
        print(f"\n   [{weights[0]}{weights[1]} / {images}] input-channel spread per layer (binades): "
              + " ".join(f"{v:.1f}" for v in spreads) + f" -> {'bf16x3' if flags else 'fp16x2'}")


@pytest.mark.parametrize("n_res,HW", [(1, 32), (3, 32), (2, 64), (2, 96)])
@pytest.mark.parametrize("kind", ["coupled", "independent"])
def test_per_layer_two_term_kernels_on_heterogeneous_channel_scales(n_res, HW, kind, capsys):
    """The same question for the kernels BEHIND the four-kernel path: with one or three residual layers the default shapes run the
    per-layer / per-pair two-term fp16 kernels (conv_tile8_bf3<., H2>, res_tile8 / res_pair8: their own per-output-channel scale
    tables and hand-over of image maxima); 64x64 / 96x96 images put 16x16 / 24x24 latent maps on the halo-tile kernels of BASELINE
    configs 4 / 5 (conv_halo8_h2, res_halo8_h2).  Encoder and decoder per output channel against the reference, and against fp64."""
    from oracle import torch_port
    from vqvae_amd import _lib, conv
    from vqvae_amd.modules import VQVAE
    conv.set_conv_backend("hip")
    h, rh, K, D = DIMS[0], DIMS[1], DIMS[3], DIMS[4]
    sd0 = torch_port.init_state_dict(h, rh, K, D, seed=0, n_res_layers=n_res)
    sd = hetero.rescale_coupled(sd0, 1, 3.0, n_res) if kind == "coupled" else hetero.rescale_independent(sd0, 1, 3.0, n_res)
    Bn = B if HW == 32 else 6
    x = torch.randn(Bn, 3, HW, HW, generator=torch.Generator().manual_seed(79))
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        z_e = torch_port.encode(sd, x.clone(), n_res)
        z_e64 = torch_port.encode(sd64, x.double(), n_res)
        _, z_q, _, _, _ = torch_port.quantize(z_e, sd["vector_quantization.embedding.weight"], 0.25)
        x_hat = torch_port.decode(sd, z_q.clone(), n_res)
        x_hat64 = torch_port.decode(sd64, z_q.double(), n_res)
    m = VQVAE(h, rh, n_res, K, D, 0.25).eval()
    m.load_state_dict(sd)
    m = m.to(dev())
    L = _lib.load()
    cw, _keep = m._c_weights()
    nws = L.vqvae_workspace_bytes(cw.dims, Bn, HW, HW)
    assert nws > 0
    ws = torch.empty(nws, dtype=torch.uint8, device=dev())
    st = torch.cuda.current_stream().cuda_stream
    xd = x.to(dev()).contiguous()
    with torch.no_grad():
        ze_d = torch.empty(Bn, HW // 4, HW // 4, D, device=dev())
        _lib.check(L.vqvae_encoder_f32(cw, xd.data_ptr(), Bn, HW, HW, ze_d.data_ptr(), ws.data_ptr(), nws, st))
        zq_rows = z_q.to(dev()).permute(0, 2, 3, 1).contiguous()
        xh_d = torch.empty_like(xd)
        _lib.check(L.vqvae_decoder_f32(cw, zq_rows.data_ptr(), Bn, HW // 4, HW // 4, xh_d.data_ptr(), ws.data_ptr(), nws, st))
    torch.cuda.synchronize()
    ze = ze_d.permute(0, 3, 1, 2).cpu().numpy()
    w_ze = hetero.per_channel_check(ze, z_e.numpy(), "z_e", per_image=False)
    # (per (image, channel) against fp64 the distance grows with depth: 2.2e-5 measured behind three residual layers where the fp32
    # reference itself is 6.6e-6 away; the bound of the two-layer suite above, 2e-5, is doubled here)
    o_ze, t_ze = _vs_fp64(ze, z_e.numpy(), z_e64.numpy(), "z_e", lim=4e-5)
    w_xh = hetero.per_channel_check(xh_d.cpu().numpy(), x_hat.numpy(), "x_hat", per_image=False)
    o_xh, t_xh = _vs_fp64(xh_d.cpu().numpy(), x_hat.numpy(), x_hat64.numpy(), "x_hat", lim=4e-5)
    with capsys.disabled():
        print(f"\n   [{kind}, {n_res} residual layer(s), {HW}x{HW} images, per-layer kernels] worst error in units of the channel maximum: z_e {w_ze:.2e}, x_hat {w_xh:.2e};"
              f"  vs fp64 per (image, channel): z_e {o_ze:.2e} (fp32 reference {t_ze:.2e}), x_hat {o_xh:.2e} ({t_xh:.2e})")
