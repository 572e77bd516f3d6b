"""hipGraph replay of the forward (vqvae_amd/graph.py): same bits as eager launches, new inputs honoured."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graph_replay_equals_eager_bitwise():
    from vqvae_amd import conv
    from vqvae_amd.graph import GraphedForward
    from vqvae_amd.modules import VQVAE
    conv.set_conv_backend("hip")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = VQVAE(128, 32, 2, 512, 64, 0.25).to(dev).eval()
    x1 = torch.randn(32, 3, 32, 32, device=dev)
    x2 = torch.randn(32, 3, 32, 32, device=dev)
    with torch.no_grad():
        e1 = [t.clone() for t in m(x1)]
        e2 = [t.clone() for t in m(x2)]
    g = GraphedForward(m, x1)
    for x, e in ((x1, e1), (x2, e2), (x1, e1)):
        out = g(x)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, e)), "graph replay must reproduce the eager bits"
    with pytest.raises(ValueError):
        g(torch.randn(16, 3, 32, 32, device=dev))
    with pytest.raises(Exception):
        GraphedForward(m, x1.cpu())


@pytest.mark.parametrize("dims,B,HW", [((128, 32, 2, 1024, 64), 16, 32),      # K = 1024: conv kernels + the streamed quantizer as separate launches
                                       ((64, 32, 1, 512, 64), 8, 32)])        # h_dim 64: per-layer kernels + the stand-alone stream-tracker quantizer
def test_graph_replay_of_the_unfused_paths(dims, B, HW):
    """Shapes outside the four-kernel path launch the quantizer on its own (vq_track_kernel_d64 by a plain launch when no profile is
    being taken -- the extended launch with dispatch events is not something a stream capture should have to record)."""
    from vqvae_amd import conv
    from vqvae_amd.graph import GraphedForward
    from vqvae_amd.modules import VQVAE
    conv.set_conv_backend("hip")
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = VQVAE(*dims, 0.25).to(dev).eval()
    x1 = torch.randn(B, 3, HW, HW, device=dev)
    x2 = torch.randn(B, 3, HW, HW, device=dev)
    with torch.no_grad():
        e1 = [t.clone() for t in m(x1)]
        e2 = [t.clone() for t in m(x2)]
    g = GraphedForward(m, x1)
    for x, e in ((x2, e2), (x1, e1)):
        out = g(x)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, e))


def test_graph_capture_of_a_sub_module_through_the_per_layer_cache():
    """Encoder / Decoder on their own go through conv_hip's per-layer packed-weight cache, whose entries carry the event
    recorded behind their pack launches.  A hit inside a stream capture must not QUERY that event (not a capturable call): the
    packing stream is recognised as such, any other stream waits on the event.  Capture right after the warm-up that packed the
    weights (their events are still attached), on the capture stream and from a cache filled on ANOTHER stream."""
    from vqvae_amd import conv
    from vqvae_amd.graph import GraphedForward
    from vqvae_amd.modules import VQVAE
    conv.set_conv_backend("hip")
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    m = VQVAE(128, 32, 2, 512, 64, 0.25).to(dev).eval()
    x1 = torch.randn(16, 3, 32, 32, device=dev)
    x2 = torch.randn(16, 3, 32, 32, device=dev)
    g = GraphedForward(m.encoder, x1, warmup=1)          # packs on the capture stream, captures at once
    with torch.no_grad():
        e1, e2 = m.encoder(x1).clone(), m.encoder(x2).clone()
    for x, e in ((x2, e2), (x1, e1)):
        out = g(x)
        torch.cuda.synchronize()
        assert torch.equal(out, e)
    # the decoder's images are packed on a side stream first; the capture (another stream) then meets entries with foreign events
    z = torch.randn(16, 64, 8, 8, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.no_grad(), torch.cuda.stream(side):
        d_ref = m.decoder(z).clone()
    g2 = GraphedForward(m.decoder, z, warmup=1)      # (its warm-up waits on the side stream's pack events; the capture itself does not)
    torch.cuda.current_stream(dev).wait_stream(side)
    out = g2(z)
    torch.cuda.synchronize()
    assert torch.equal(out, d_ref)
