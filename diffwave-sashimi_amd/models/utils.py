"""Host-side helpers shared by the model shims."""
import math

import torch
import torch.nn as nn


def calc_diffusion_step_embedding(diffusion_steps, diffusion_step_embed_dim_in):
    """Same contract as the reference helper (``models/utils.py:4-29``) but
    device-agnostic (the reference hard-codes ``.cuda()`` at ``:24``).  Kept for
    API completeness; the engine evaluates the embedding inside libdws.so."""
    assert diffusion_step_embed_dim_in % 2 == 0
    half = diffusion_step_embed_dim_in // 2
    freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(diffusion_steps.device)
    arg = diffusion_steps * freq
    return torch.cat((torch.sin(arg), torch.cos(arg)), 1)


def _uniform_(t, fan_in):
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
    with torch.no_grad():
        return t.uniform_(-bound, bound)


class WNParams(nn.Module):
    """Parameter holder of a weight-normed conv: ``bias``, ``weight_g``,
    ``weight_v`` exactly as ``torch.nn.utils.weight_norm`` registers them
    (``models/wavenet.py:20-21``).  Initialised like the reference: torch's
    default conv init for ``v``/``bias`` and ``g = ||v||`` (the
    ``kaiming_normal_`` at ``wavenet.py:22`` touches only the derived
    ``.weight`` attribute).  ``norm_dims`` are the dims the
    norm runs over (all but 0)."""

    def __init__(self, v_shape, fan_in, n_bias):
        super().__init__()
        v = _uniform_(torch.empty(*v_shape), fan_in)
        self.bias = nn.Parameter(_uniform_(torch.empty(n_bias), fan_in))
        g_shape = [v_shape[0]] + [1] * (len(v_shape) - 1)
        norm = torch.linalg.vector_norm(v.reshape(v_shape[0], -1), 2, 1).reshape(g_shape)
        self.weight_g = nn.Parameter(norm)
        self.weight_v = nn.Parameter(v)


class ConvParams(nn.Module):
    """``Conv`` of the reference (``models/wavenet.py:16-26``): a module whose
    ``.conv`` child is the weight-normed Conv1d."""

    def __init__(self, in_channels, out_channels, kernel_size):
        super().__init__()
        self.conv = WNParams((out_channels, in_channels, kernel_size), in_channels * kernel_size, out_channels)


class ZeroConvParams(nn.Module):
    """``ZeroConv1d`` (``models/wavenet.py:31-40``): plain 1x1 conv, zero-initialised."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Module()
        self.conv.weight = nn.Parameter(torch.zeros(out_channels, in_channels, 1))
        self.conv.bias = nn.Parameter(torch.zeros(out_channels))


class LinearParams(nn.Module):
    """``nn.Linear`` parameter holder with torch's default init."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(_uniform_(torch.empty(out_features, in_features), in_features))
        self.bias = nn.Parameter(_uniform_(torch.empty(out_features), in_features))


def upsampler_params(mel_upsample):
    """Two weight-normed ``ConvTranspose2d(1,1,(3,2s))`` (``models/wavenet.py:64-69``);
    ConvTranspose weight layout is [in, out, kh, kw] so the dim-0 norm spans the whole kernel."""
    ml = nn.ModuleList()
    for s in mel_upsample:
        ml.append(WNParams((1, 1, 3, 2 * s), 1 * 3 * 2 * s, 1))
    return ml
