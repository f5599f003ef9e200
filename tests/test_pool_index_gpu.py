"""GPU: the DownPool / UpPool index maps (`sashimi.py:37,57`) are integer-exact on the HIP path.

north_star: "bit-exact for the pooling index path".  The pools sit inside the network, so the test builds a network
around them that is exact in fp32: snet (no blocks on the way down), init conv `x[b,h,l] = w * audio[b,l] + bias[h]`
producing small integers, IDENTITY pool weights (weight_v = I, weight_g = 1 -> the folded weight is exactly I), and a
centre block made transparent (output_linear = 0 and ff.2 weight_g = 0 -> the block returns `x + skip = 2x` exactly).
Then, read through the engine's activation taps,

    out:d_layers.0 = rearrange(x, 'b h (l s) -> b (h s) l')          (DownPool, compared with the reference fixture)
    out:u_layers.0 = rearrange(c, 'b (h s) l -> b h (l s)') + x      (UpPool + the skip it adds, `sashimi.py:303-306`)

must hold with torch.equal.  The fixtures `pool/*` of tests/golden/s4_parts.npz come from einops' rearrange as the
reference calls it; the small case feeds exactly those tensors, the MFMA-sized case (`pw_mfma_kernel`) the same maps on a
larger arange."""
import pytest
import torch

from tests import cases
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _down(x, p):
    B, H, L = x.shape
    return x.reshape(B, H, L // p, p).permute(0, 1, 3, 2).reshape(B, H * p, L // p)


def _up(y, p):
    B, HP, L = y.shape
    return y.reshape(B, HP // p, p, L).permute(0, 1, 3, 2).reshape(B, HP // p, L * p)


def _exact_net(gpu, H, L, p, w_init, bias_step):
    cfg = cases.ss_cfg(d_model=H, n_layers=1, L=L, pool=[p], expand=p, unet=False, diffusion_step_embed_dim_mid=64)
    net = cases.build_ours(cfg, 5).to(gpu)
    sd = net.state_dict()
    eye = torch.eye(H * p, device=gpu).reshape(H * p, H * p, 1)
    with torch.no_grad():
        sd["init_conv.0.conv.weight_v"].fill_(1.0)
        sd["init_conv.0.conv.weight_g"].fill_(w_init)
        sd["init_conv.0.conv.bias"].copy_(torch.arange(H, device=gpu, dtype=torch.float32) * bias_step)
        for pre in ("d_layers.0", "u_layers.0"):
            sd[pre + ".linear.conv.weight_v"].copy_(eye)
            sd[pre + ".linear.conv.weight_g"].fill_(1.0)
            sd[pre + ".linear.conv.bias"].zero_()
        sd["c_layers.0.layer.output_linear.0.weight"].zero_()
        sd["c_layers.0.layer.output_linear.0.bias"].zero_()
        sd["c_layers.0.ff.ff.2.conv.weight_g"].zero_()
        sd["c_layers.0.ff.ff.2.conv.bias"].zero_()
    return net


def _taps(net, gpu, audio, H, L, p):
    B = audio.shape[0]
    with torch.no_grad():
        net((audio.to(gpu), torch.zeros(B, 1, device=gpu)))
    torch.cuda.synchronize()
    d = net.read_tap("out:d_layers.0", (B, H * p, L // p)).cpu()
    c = net.read_tap("out:c_layers.0", (B, H * p, L // p)).cpu()
    u = net.read_tap("out:u_layers.0", (B, H, L)).cpu()
    return d, c, u


def test_pool_index_maps_equal_the_reference_fixtures(gpu):
    """Generic kernels (H = 3): the fixtures themselves go through the HIP pools."""
    g = load_golden("s4_parts")
    x = torch.from_numpy(g["pool/x"])                 # arange [2, 3, 20]
    B, H, L = x.shape
    # DownPool: x_init == pool/x  (audio[b, l] = b*H*L + l, bias[h] = h*L, weight 1)
    net = _exact_net(gpu, H, L, 4, 1.0, float(L))
    audio = (torch.arange(B).view(B, 1, 1) * (H * L) + torch.arange(L).view(1, 1, L)).float()
    d, c, u = _taps(net, gpu, audio, H, L, 4)
    assert torch.equal(d, torch.from_numpy(g["pool/down_p4"]))
    assert torch.equal(c, 2 * d) and torch.equal(u, 3 * x)          # up(down(x)) = x, + skip x, centre doubled
    # UpPool: make the centre output == pool/y, i.e. x_init = up_p4 / 2 (weight 1/2, bias 10 h, audio = 60 b + f(i))
    y, up = torch.from_numpy(g["pool/y"]), torch.from_numpy(g["pool/up_p4"])
    net = _exact_net(gpu, H, L, 4, 0.5, 10.0)
    i = torch.arange(L)
    audio = (torch.arange(B).view(B, 1, 1) * 60 + (5 * (i % 4) + i // 4).view(1, 1, L)).float()
    d, c, u = _taps(net, gpu, audio, H, L, 4)
    assert torch.equal(c, y)                                         # the UpPool's input IS the fixture's input
    assert torch.equal(u, up + up / 2)                               # rearranged fixture output + the skip (= up/2)


@pytest.mark.parametrize("H,L,p", [(64, 1024, 4), (32, 512, 4), (64, 512, 2)])
def test_pool_index_maps_exact_on_the_mfma_kernels(gpu, H, L, p):
    """Channel counts of the real configs: `pw_mfma_kernel` folds the maps into its B-operand gather / float4
    scatter (or `pool_rearrange` + GEMM for shapes it does not tile); same exactness, larger arange."""
    B = 2
    net = _exact_net(gpu, H, L, p, 1.0, float(L))
    audio = (torch.arange(B).view(B, 1, 1) * (H * L) + torch.arange(L).view(1, 1, L)).float()
    x = torch.arange(B * H * L, dtype=torch.float32).reshape(B, H, L)
    d, c, u = _taps(net, gpu, audio, H, L, p)
    assert torch.equal(d, _down(x, p))
    assert torch.equal(c, 2 * d)
    assert torch.equal(u, _up(c, p) + x)
