"""Case table shared by make_golden.py (generator, needs the reference checkout) and the tests."""
from tests import cases

GRAD_CASES = {
    "wn": (cases.wn_cfg(res_channels=16, skip_channels=16, num_res_layers=4, dilation_cycle=4,
                        diffusion_step_embed_dim_mid=32, diffusion_step_embed_dim_out=32), 2, 96, None),
    "wn_cond": (cases.wn_cfg(unconditional=False, res_channels=16, skip_channels=16, num_res_layers=2, dilation_cycle=2,
                             mel_upsample=[16, 16], diffusion_step_embed_dim_mid=32, diffusion_step_embed_dim_out=32), 2, 256, 1),
    "ss": (cases.ss_cfg(d_model=8, n_layers=1, L=256, diffusion_step_embed_dim_mid=16), 2, 256, None),
    "ss_cond": (cases.ss_cfg(unconditional=False, d_model=8, n_layers=1, L=256, pool=[4], mel_upsample=[16, 16],
                             diffusion_step_embed_dim_mid=16), 2, 256, 1),
}

# Channel counts the engine's MFMA adjoints train at (H = 32 / 64 / 128): stored in grads_d32.npz, every gradient
# tensor subsampled to <= GRAD_KEEP entries (``grad_slice``) so the fixture stays small.
GRAD_CASES_D32 = {
    "ss_d32": (cases.ss_cfg(d_model=32, n_layers=1, L=1024, diffusion_step_embed_dim_mid=16), 2, 1024, None),
}
GRAD_KEEP = 4096


def grad_slice(t):
    """The entries of a gradient tensor kept in the subsampled fixture (flattened, fixed stride)."""
    f = t.reshape(-1)
    return f[::max(1, -(-f.numel() // GRAD_KEEP))]
