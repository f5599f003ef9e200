"""BASELINE.json's workloads, the hardware peaks the rooflines are priced against, and the seeded synthetic model."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {
    # BASELINE.json configs[1]
    "wnet_h256_d36_T200": dict(
        model=dict(_name_="wavenet", unconditional=True, in_channels=1, out_channels=1,
                   diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                   diffusion_step_embed_dim_out=512, res_channels=256, skip_channels=256,
                   num_res_layers=36, dilation_cycle=12),
        diffusion=dict(T=200, beta_0=1e-4, beta_T=0.02), B=16, L=16000),
    # BASELINE.json configs[0] (the reference's CPU-runnable case)
    "wnet_h128_d30_T200": dict(
        model=dict(_name_="wavenet", unconditional=True, in_channels=1, out_channels=1,
                   diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                   diffusion_step_embed_dim_out=512, res_channels=128, skip_channels=256,
                   num_res_layers=30, dilation_cycle=10),
        diffusion=dict(T=200, beta_0=1e-4, beta_T=0.02), B=16, L=16000),
    # BASELINE.json configs[2]
    "unet_d64_n6_T200": dict(
        model=dict(_name_="sashimi", unconditional=True, in_channels=1, out_channels=1,
                   diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                   diffusion_step_embed_dim_out=512, unet=True, d_model=64, n_layers=6, pool=[4, 4],
                   expand=2, ff=2, L=16000),
        diffusion=dict(T=200, beta_0=1e-4, beta_T=0.02), B=16, L=16000),
    # sampling with the architecture of BASELINE.json configs[4] (unet_d128_n6; README.md:215 samples it at B=128/GPU)
    "unet_d128_n6_T200": dict(
        model=dict(_name_="sashimi", unconditional=True, in_channels=1, out_channels=1,
                   diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                   diffusion_step_embed_dim_out=512, unet=True, d_model=128, n_layers=6, pool=[4, 4],
                   expand=2, ff=2, L=16000),
        diffusion=dict(T=200, beta_0=1e-4, beta_T=0.02), B=16, L=16000),
    # BASELINE.json configs[3] (mel conditioner installed once per utterance)
    "unet_d32_n6_T50_cond": dict(
        model=dict(_name_="sashimi", unconditional=False, in_channels=1, out_channels=1,
                   diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                   diffusion_step_embed_dim_out=512, unet=True, d_model=32, n_layers=6, pool=[4, 4],
                   expand=2, ff=2, L=16000, mel_upsample=[16, 16]),
        diffusion=dict(T=50, beta_0=1e-4, beta_T=0.05), B=32, L=16000, Tmel=63),
}

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA (three bf16 MFMAs per fp32-equivalent product)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak


def build_model(cfg, device):
    """Random-init weights of the named architecture (no checkpoints exist offline):
    reference initialisers under manual_seed(0), final zero-conv re-initialised
    N(0, 0.1^2) so the network output is not identically zero."""
    from diffwave_sashimi_amd.models import construct_model
    torch.manual_seed(0)
    net = construct_model(dict(cfg["model"]))
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        sd = net.state_dict()
        for k in ("final_conv.2.conv.weight", "final_conv.2.conv.bias"):
            sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.1)
    return net.to(device).eval()



DTYPE_NAMES = {
    "f32": "f32",
    "bf16x3": "bf16x3 split (hi/lo bf16 MFMA inputs, fp32 accumulate; ~1e-5 rel)",
    "bf16x6": "f32-equivalent (3-term bf16 split, 6 products, fp32 accumulate)",
    "f16x3": "f32-class (2-term fp16 split of power-of-two scaled operands, 3 products, fp32 accumulate; 22 bits per operand)",
}


