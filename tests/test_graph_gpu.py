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
