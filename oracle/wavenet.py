"""Oracle: WaveNet backbone (``models/wavenet.py``) as functional torch-CPU ops
over a reference-layout ``state_dict``.  Test infrastructure only.

The op order (conv1d / linear / elementwise) is kept identical to the
reference so that this file also serves as the "reference-equivalent" CPU
baseline timed by ``bench.py`` (no hoisting, weight-norm re-evaluated per
call, mel conditioner recomputed in every block).
"""
import math

import torch
import torch.nn.functional as F

from .diffusion import calc_diffusion_step_embedding


def swish(x):
    """``models/wavenet.py:10-11``."""
    return x * torch.sigmoid(x)


def weight_norm_weight(sd, prefix):
    """``torch.nn.utils.weight_norm`` with the default ``dim=0`` as applied in
    ``models/wavenet.py:21``: ``W[o] = g[o] * v[o] / ||v[o]||_2`` with the norm
    over every dim except 0.  For the ``ConvTranspose2d(1,1,...)`` upsamplers
    (``wavenet.py:66-67``) dim 0 has extent 1, i.e. one norm over the whole
    kernel."""
    g = sd[prefix + ".weight_g"]
    v = sd[prefix + ".weight_v"]
    dims = tuple(range(1, v.dim()))
    return v * (g / torch.linalg.vector_norm(v, 2, dims, keepdim=True))


def wn_conv1d(sd, prefix, x, dilation=1):
    """``Conv.forward`` (``models/wavenet.py:16-26``): weight-normed Conv1d with
    ``padding = dilation*(k-1)//2``."""
    w = weight_norm_weight(sd, prefix)
    k = w.shape[-1]
    return F.conv1d(x, w, sd[prefix + ".bias"], dilation=dilation, padding=dilation * (k - 1) // 2)


def mel_upsample(sd, prefix, mel_spec, L):
    """Mel conditioner front half (``models/wavenet.py:98-108`` ==
    ``models/sashimi.py:160-172``): two weight-normed ConvTranspose2d +
    leaky_relu(0.4), then truncation to the first ``L`` frames."""
    m = torch.unsqueeze(mel_spec, dim=1)
    for i in range(2):
        p = f"{prefix}.upsample_conv2d.{i}"
        w = weight_norm_weight(sd, p)
        s = w.shape[-1] // 2
        m = F.conv_transpose2d(m, w, sd[p + ".bias"], stride=(1, s), padding=(1, s // 2))
        m = F.leaky_relu(m, 0.4)
    m = torch.squeeze(m, dim=1)
    assert m.size(2) >= L
    if m.size(2) > L:
        m = m[:, :, :L]
    return m


def residual_block(sd, prefix, x, diffusion_step_embed, dilation, mel_spec=None):
    """``Residual_block.forward`` (``models/wavenet.py:82-121``)."""
    B, C, L = x.shape
    part_t = F.linear(diffusion_step_embed, sd[prefix + ".fc_t.weight"], sd[prefix + ".fc_t.bias"])
    h = x + part_t.view([B, C, 1])
    h = wn_conv1d(sd, prefix + ".dilated_conv_layer.conv", h, dilation=dilation)
    if mel_spec is not None:
        m = mel_upsample(sd, prefix, mel_spec, L)
        h = h + wn_conv1d(sd, prefix + ".mel_conv.conv", m)
    out = torch.tanh(h[:, :C, :]) * torch.sigmoid(h[:, C:, :])
    res = wn_conv1d(sd, prefix + ".res_conv", out)
    skip = wn_conv1d(sd, prefix + ".skip_conv", out)
    return (x + res) * math.sqrt(0.5), skip


def step_embedding_mlp(sd, prefix, diffusion_steps, dim_in=128):
    """Shared two-layer MLP with swish (``models/wavenet.py:153-155`` ==
    ``models/sashimi.py:287-289``).  ``prefix`` includes the trailing dot (or
    is empty for SaShiMi where fc_t1/fc_t2 hang off the root module)."""
    e = calc_diffusion_step_embedding(diffusion_steps, dim_in)
    e = e.to(sd[prefix + "fc_t1.weight"].dtype)   # float64 state_dict: the same graph in double (conditioning studies)
    e = swish(F.linear(e, sd[prefix + "fc_t1.weight"], sd[prefix + "fc_t1.bias"]))
    e = swish(F.linear(e, sd[prefix + "fc_t2.weight"], sd[prefix + "fc_t2.bias"]))
    return e


def wavenet_forward(sd, cfg, audio, diffusion_steps, mel_spec=None, return_pre_final=False):
    """``WaveNet.forward`` (``models/wavenet.py:202-210``) incl.
    ``Residual_group.forward`` (``:149-165``)."""
    n_layers = cfg["num_res_layers"]
    cycle = cfg["dilation_cycle"]
    x = F.relu(wn_conv1d(sd, "init_conv.0.conv", audio))
    emb = step_embedding_mlp(sd, "residual_layer.", diffusion_steps, cfg.get("diffusion_step_embed_dim_in", 128))
    h = x
    skip = 0
    for n in range(n_layers):
        h, skip_n = residual_block(sd, f"residual_layer.residual_blocks.{n}", h, emb,
                                   2 ** (n % cycle), mel_spec=mel_spec)
        skip = skip + skip_n
    skip = skip * math.sqrt(1.0 / n_layers)
    y = F.relu(wn_conv1d(sd, "final_conv.0.conv", skip))
    out = F.conv1d(y, sd["final_conv.2.conv.weight"], sd["final_conv.2.conv.bias"])
    if return_pre_final:
        return out, y
    return out


class WaveNetOracle:
    """Callable with the reference's ``net((audio, t), mel_spec=None)`` surface."""

    def __init__(self, sd, cfg):
        self.sd = {k: v.detach().float() if v.is_floating_point() else v for k, v in sd.items()}
        self.cfg = dict(cfg)

    def __call__(self, input_data, mel_spec=None):
        audio, steps = input_data
        with torch.no_grad():
            return wavenet_forward(self.sd, self.cfg, audio, steps, mel_spec=mel_spec)
