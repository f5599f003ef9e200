"""Shared definitions of the parity cases: model configs, seeds and input
generators.  Used by ``tests/golden/make_golden.py`` (build container, with the
reference imported) and by the tests (anywhere) so both sides build the very
same tensors from the same seeds."""
import torch

WN_BASE = dict(_name_="wavenet", unconditional=True, in_channels=1, out_channels=1,
               diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
               diffusion_step_embed_dim_out=512)


def wn_cfg(**kw):
    c = dict(WN_BASE)
    c.update(kw)
    return c


# name -> (cfg, B, L, weight_seed, input_seed, store_weights)
WAVENET_CASES = {
    # generic (non-MFMA) kernels, dilations 1..1024 with L=300: taps fully out of range
    "wn_tiny": (wn_cfg(res_channels=16, skip_channels=16, num_res_layers=11, dilation_cycle=11), 2, 300, 11, 12, True),
    # MFMA kernels, C=S=64, dilations 1..2048, L not a multiple of the 64-position tile
    "wn_c64": (wn_cfg(res_channels=64, skip_channels=64, num_res_layers=12, dilation_cycle=12), 2, 600, 21, 22, False),
    # the reference's small config (configs/model/wavenet_small.yaml), shortened stack
    "wn_c128": (wn_cfg(res_channels=128, skip_channels=256, num_res_layers=10, dilation_cycle=10), 1, 2048, 31, 32, False),
    # BASELINE config 1: wnet_h128_d30, B=1, L=16000
    "wn_h128_d30": (wn_cfg(res_channels=128, skip_channels=256, num_res_layers=30, dilation_cycle=10), 1, 16000, 41, 42, False),
    # BASELINE config 2 architecture (wnet_h256_d36) at B=1, L=4096
    "wn_h256_d36": (wn_cfg(res_channels=256, skip_channels=256, num_res_layers=36, dilation_cycle=12), 1, 4096, 51, 52, False),
}

# conditional WaveNet: mel [Bm, 80, Tmel] -> L = Tmel*256 (hop 256 = 16*16)
WAVENET_COND_CASES = {
    "wn_cond_tiny": (wn_cfg(unconditional=False, res_channels=16, skip_channels=16, num_res_layers=3,
                            dilation_cycle=3, mel_upsample=[16, 16]), 2, 500, 2, 61, 62, True),
    "wn_cond_c64": (wn_cfg(unconditional=False, res_channels=64, skip_channels=64, num_res_layers=3,
                           dilation_cycle=3, mel_upsample=[16, 16]), 2, 512, 2, 71, 72, False),
}


SS_BASE = dict(_name_="sashimi", unconditional=True, in_channels=1, out_channels=1,
               diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
               diffusion_step_embed_dim_out=512, unet=True, pool=[4, 4], expand=2, ff=2)


def ss_cfg(**kw):
    c = dict(SS_BASE)
    c.update(kw)
    return c


# name -> (cfg, B, weight_seed, input_seed, store_weights); input length is cfg["L"]
SASHIMI_CASES = {
    # generic (non-MFMA) kernels; small embedding MLP so the stored state_dict stays small
    "ss_tiny": (ss_cfg(d_model=8, n_layers=2, L=1024, diffusion_step_embed_dim_mid=64), 2, 111, 112, True),
    # snet variant: no blocks on the down path, skip-add only after UpPool (`sashimi.py:243,306`)
    "ss_snet": (ss_cfg(d_model=8, n_layers=2, L=256, unet=False, diffusion_step_embed_dim_mid=64), 2, 121, 122, True),
    # one pooling stage, pool 2, expand 3, ff 1: the non-default knobs
    "ss_knobs": (ss_cfg(d_model=6, n_layers=1, L=250, pool=[2], expand=3, ff=1, diffusion_step_embed_dim_mid=64), 3, 131, 132, True),
    # channel counts of BASELINE config 3 (H = 64/128/256), shortened
    "ss_d64_short": (ss_cfg(d_model=64, n_layers=2, L=1024), 2, 141, 142, False),
    # channel counts of BASELINE config 5 (unet_d128: H = 128/256/512), shortened
    "ss_d128_short": (ss_cfg(d_model=128, n_layers=1, L=1024), 1, 181, 182, False),
    # BASELINE config 3 architecture: unet_d64_n6 pool[4,4] ff2, L=16000, at B=1
    "ss_unet_d64": (ss_cfg(d_model=64, n_layers=6, L=16000), 1, 151, 152, False),
}

# conditional SaShiMi: name -> (cfg, B, Tmel, weight_seed, input_seed, store)
SASHIMI_COND_CASES = {
    "ss_cond_tiny": (ss_cfg(unconditional=False, d_model=8, n_layers=1, L=512, mel_upsample=[16, 16],
                            diffusion_step_embed_dim_mid=64), 2, 2, 161, 162, True),
    # BASELINE config 4 channel counts (unet_d32), shortened: L=1024 = 4 mel frames * hop 256
    "ss_cond_d32": (ss_cfg(unconditional=False, d_model=32, n_layers=2, L=1024, mel_upsample=[16, 16]), 2, 4, 171, 172, False),
}


# BASELINE config 4 at its own geometry: unet_d32_n6, mel-conditional, L = 16000 = 62.5 frames of hop 256 -> mel
# [., 80, 63] upsampled to 16128 and truncated to 16000 / 4000 / 1000 per stage (`sashimi.py:160-175`), T = 50.
# (cfg, B, Tmel, weight_seed, input_seed)
SASHIMI_C4 = (ss_cfg(unconditional=False, d_model=32, n_layers=6, L=16000, mel_upsample=[16, 16]), 2, 63, 191, 192)


def randomize_zero_conv(model, seed):
    """``final_conv[2]`` is zero-initialised in the reference (`wavenet.py:35-36`)
    so an untrained net outputs 0; re-initialise it N(0, 0.1^2) or parity is
    vacuous (semantic trap 2)."""
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    with torch.no_grad():
        for k in ("final_conv.2.conv.weight", "final_conv.2.conv.bias"):
            sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.1)


def build_ours(cfg, weight_seed):
    """Our parameter-holder module, seeded; this is where synthetic weights come from."""
    from diffwave_sashimi_amd.models import construct_model
    torch.manual_seed(weight_seed)
    m = construct_model(dict(cfg))
    randomize_zero_conv(m, weight_seed + 1000)
    return m.eval()


def wavenet_inputs(B, L, in_channels, input_seed, T=200):
    g = torch.Generator().manual_seed(input_seed)
    audio = torch.randn(B, in_channels, L, generator=g)
    steps = torch.randint(0, T, (B, 1), generator=g).float()
    return audio, steps


def mel_inputs(Bm, Tmel, input_seed):
    """log-mel range of `dataloaders/stft.py:84-90`: U(-11.5, 2)."""
    g = torch.Generator().manual_seed(input_seed + 7)
    return torch.rand(Bm, 80, Tmel, generator=g) * 13.5 - 11.5


def summarize(t, stride=16, edge=256):
    """Small, size-independent digest of a big output tensor."""
    f = t.detach().double().flatten()
    return dict(first=t.detach().flatten()[:edge].clone(), last=t.detach().flatten()[-edge:].clone(),
                strided=t.detach().flatten()[::stride].clone(),
                sum=f.sum().reshape(1), sumsq=(f * f).sum().reshape(1), absmax=f.abs().max().reshape(1))


# Results of expensive CPU-oracle evaluations shared between test modules of one pytest session (the float64 forward of a
# seeded case is the yardstick of several precision tests): key -> value, computed once.
_SESSION_CACHE = {}


def cached(key, make):
    if key not in _SESSION_CACHE:
        _SESSION_CACHE[key] = make()
    return _SESSION_CACHE[key]
