"""Function-level GPU parity for the small functions that whole-network tests only see through their sum
(SURVEY.md section 8, rows a1, a2, a7, a9): each is read out of the HIP engine through an activation tap
(`dws_model_read_tap`) and compared with the reference's own fixture or with the oracle.

  a1  calc_diffusion_step_embedding (`models/utils.py:20-27`)    tap "emb"      vs tests/golden/embedding.npz
  a2  embedding MLP + every block's fc_t (`wavenet.py:153-155,89`, `sashimi.py:287-289,151`)
                                                                 taps "emb_mlp", "part_t" vs the oracle
  a7  TransposedLN (`models/sashimi.py:17-20`)                   tap "nfin"     vs s4_parts.npz ln/*  (the fixture's
                                                                 tensors go through the HIP LayerNorm)
  a9  FF (`models/sashimi.py:60-75`)                             taps "out:<block>" of a block whose S4 branch is
                                                                 switched off vs oracle.sashimi.ff(LN2(x))
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sashimi as oss
from oracle import wavenet as own
from tests import cases
from tests.conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _ulps(a, b):
    """max |a - b| in units of the fp32 spacing at b (values in [-1, 1]: spacing of the binade below 1 as the floor)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    sp = np.spacing(np.maximum(np.abs(b), np.float32(2.0 ** -3)))
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)) / sp))


@pytest.mark.parametrize("backbone", ["wavenet", "sashimi"])
def test_step_embedding_matches_the_reference_fixture(gpu, backbone):
    g = load_golden("embedding")
    for key_t, key_e, dim in (("t_float", "emb_float", 128), ("t_int", "emb_int", 128), ("t_float", "emb_float_64", 64)):
        if backbone == "wavenet":
            cfg = dict(cases.WAVENET_CASES["wn_tiny"][0]); cfg["diffusion_step_embed_dim_in"] = dim
            L = 64
        else:
            cfg = dict(cases.SASHIMI_CASES["ss_tiny"][0]); cfg["diffusion_step_embed_dim_in"] = dim
            L = cfg["L"]
        net = cases.build_ours(cfg, 3).to(gpu)
        t = torch.from_numpy(g[key_t])                     # [5, 1], float32 or int64 (`train.py:218`)
        B = t.shape[0]
        with torch.no_grad():
            net((torch.zeros(B, 1, L, device=gpu), t.to(gpu)))
        emb = net.read_tap("emb", (B, dim)).cpu().numpy()
        ref = g[key_e]
        u = _ulps(emb, ref)
        print(f"{backbone} {key_e}: max abs {np.abs(emb - ref).max():.3e}, {u:.2f} ulp (of max(|ref|, 1/8))")
        # full-precision sinf / cosf on the same fp32 argument t * f_i: within a few ulps of torch's
        assert np.abs(emb - ref).max() <= 5e-7 and u <= 4.0, (key_e, u)


@pytest.mark.parametrize("backbone", ["wavenet", "sashimi"])
def test_embedding_mlp_and_fc_t_rows_match_the_oracle(gpu, backbone):
    if backbone == "wavenet":
        cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_c64"]
        L = 256
    else:
        cfg, B, wseed, iseed, _ = cases.SASHIMI_CASES["ss_d64_short"]
        L = cfg["L"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    steps = torch.tensor([[0.0], [1.0], [57.0], [199.0]])
    B = steps.shape[0]
    with torch.no_grad():
        net((torch.zeros(B, 1, L, device=gpu), steps.to(gpu)))
    Eout = cfg["diffusion_step_embed_dim_out"]
    got = net.read_tap("emb_mlp", (B, Eout)).cpu()
    prefix = "residual_layer." if backbone == "wavenet" else ""
    with torch.no_grad():
        ref = own.step_embedding_mlp(sd, prefix, steps, cfg["diffusion_step_embed_dim_in"])
    assert rel_err(got, ref) < 1e-5, rel_err(got, ref)
    # every block's fc_t(e) (`wavenet.py:89`, `sashimi.py:151`), stacked in execution order
    names = [k[: -len(".fc_t.weight")] for k in sd if k.endswith(".fc_t.weight")]
    with torch.no_grad():
        rows = {n: F.linear(ref, sd[n + ".fc_t.weight"], sd[n + ".fc_t.bias"]) for n in names}
    width = sum(r.shape[1] for r in rows.values())
    pt = net.read_tap("part_t", (B, width)).cpu()
    # the stacking order is the engine's business: every reference row must appear as a contiguous slice
    found = 0
    for n, r in rows.items():
        w = r.shape[1]
        hit = [o for o in range(0, width - w + 1) if rel_err(pt[:, o:o + w], r) < 1e-5]
        assert hit, f"fc_t row of {n} not found in part_t"
        found += 1
    assert found == len(names) and len(names) > 0
    print(f"{backbone}: emb_mlp rel err {rel_err(got, ref):.2e}; {found} fc_t rows located in part_t")


@pytest.mark.parametrize("precision", ["f32", "bf16x6"])
def test_transposed_layernorm_on_the_reference_fixture(gpu, precision):
    """(Under `precision="bf16x6"` too: whatever arithmetic the option selects for the block around it, the LayerNorm the
    fixture goes through must stay the fixture's.)  ln/x [2, 6, 40] -> HIP LayerNorm -> ln/y.  The fixture enters as a 6-channel "audio" through an identity init
    conv scaled by 1/2; its bias shifts the tensor above the ReLU (LayerNorm is shift-invariant down the channel column),
    the only block is switched off (output_linear = 0, ff.2 = 0: it returns its input, and `sashimi.py:301` adds the
    input once more: 2 (x/2 + 8) = x + 16), and `norm` carries the fixture's (m, s)."""
    g = load_golden("s4_parts")
    x, y, (m, s) = torch.from_numpy(g["ln/x"]), torch.from_numpy(g["ln/y"]), g["ln/ms"]
    B, H, L = x.shape
    cfg = cases.ss_cfg(d_model=H, in_channels=H, n_layers=1, L=L, pool=[], unet=False, diffusion_step_embed_dim_mid=64)
    net = cases.build_ours(cfg, 5).to(gpu)
    net.set_option("precision", precision)
    sd = net.state_dict()
    shift = 16.0
    assert float(x.min()) > -shift
    with torch.no_grad():
        sd["init_conv.0.conv.weight_v"].copy_(torch.eye(H, device=gpu).reshape(H, H, 1))
        sd["init_conv.0.conv.weight_g"].fill_(0.5)
        sd["init_conv.0.conv.bias"].fill_(shift / 2)
        sd["c_layers.0.layer.output_linear.0.weight"].zero_()
        sd["c_layers.0.layer.output_linear.0.bias"].zero_()
        sd["c_layers.0.ff.ff.2.conv.weight_g"].zero_()
        sd["c_layers.0.ff.ff.2.conv.bias"].zero_()
        sd["norm.m"].fill_(float(m))
        sd["norm.s"].fill_(float(s))
        net.invalidate()
        net((x.to(gpu), torch.zeros(B, 1, device=gpu)))
    out = net.read_tap("out:c_layers.0", (B, H, L)).cpu()
    assert float((out - (x + shift)).abs().max()) < 4e-6    # the block really is transparent: LN sees the fixture + 16
    got = net.read_tap("nfin", (B, H, L)).cpu()
    err = float((got - y).abs().max())
    print(f"TransposedLN fixture through the HIP kernel: max abs err {err:.2e} (|y| max {float(y.abs().max()):.2f})")
    assert err < 2e-5
    assert rel_err(got, oss.transposed_ln(x + shift, torch.tensor(float(m)), torch.tensor(float(s)))) < 2e-6


@pytest.mark.parametrize("precision", ["f32", "bf16x6"])
@pytest.mark.parametrize("H", [8, 32, 64, 128, 256])
def test_ff_branch_matches_the_oracle(gpu, H, precision):
    """(`precision="bf16x6"`: the same FF through the split tails -- `s4_tail_chain6_kernel` at H = 32 / 64,
    `s4_tail_wide6_kernel` at 128, `gemm_slab_split` at 256; H = 8 has no split instance and stays on the generic kernel.)
    out = x1 + FF(LN2(x1)) (`sashimi.py:179-184`) with the S4 branch off (x1 = x), plus the input once more
    (`sashimi.py:301`, fused into the tail): out - 2x is FF(LN2(x)) as the HIP tail computes it -- generic kernels
    (H = 8), the register-chained tail (32, 64), the LDS-tile tail (128)."""
    L, B = 512, 2
    cfg = cases.ss_cfg(d_model=H, n_layers=1, L=L, pool=[], unet=False, diffusion_step_embed_dim_mid=64)
    net = cases.build_ours(cfg, 41 + H).to(gpu)
    net.set_option("precision", precision)
    sd = net.state_dict()
    with torch.no_grad():
        sd["c_layers.0.layer.output_linear.0.weight"].zero_()
        sd["c_layers.0.layer.output_linear.0.bias"].zero_()
        net.invalidate()
    audio, steps = cases.wavenet_inputs(B, L, 1, 77)
    with torch.no_grad():
        net((audio.to(gpu), steps.to(gpu)))
    out = net.read_tap("out:c_layers.0", (B, H, L)).cpu()
    sdc = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        x = F.relu(own.wn_conv1d(sdc, "init_conv.0.conv", audio))
        ref = oss.ff(sdc, "c_layers.0.ff", oss.transposed_ln(x, sdc["c_layers.0.norm2.m"], sdc["c_layers.0.norm2.s"]))
    got = out - 2 * x
    err = rel_err(got, ref)
    print(f"H={H} {precision}: FF branch rel err {err:.2e} (|FF| max {float(ref.abs().max()):.3f}, |x| max {float(x.abs().max()):.3f})")
    assert err < 2e-5, err
