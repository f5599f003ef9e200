"""Drop-in for ``models/wavenet.py`` of the reference: same constructor kwargs,
``forward((audio, diffusion_steps), mel_spec=None)``, and state_dict keys
(``init_conv.0.conv.{bias,weight_g,weight_v}``, ``residual_layer.fc_t{1,2}.*``,
``residual_layer.residual_blocks.N.{fc_t,dilated_conv_layer.conv,res_conv,skip_conv,
upsample_conv2d.{0,1},mel_conv.conv}.*``, ``final_conv.{0,2}.conv.*``).  The
module only holds parameters; the forward runs in libdws.so."""
import torch.nn as nn

from .. import _lib
from .engine import EngineModule
from .utils import ConvParams, LinearParams, WNParams, ZeroConvParams, upsampler_params


class _ResidualBlockParams(nn.Module):
    """Parameters of ``Residual_block`` (``models/wavenet.py:45-80``)."""

    def __init__(self, res_channels, skip_channels, embed_out, unconditional, mel_upsample):
        super().__init__()
        self.fc_t = LinearParams(embed_out, res_channels)
        self.dilated_conv_layer = ConvParams(res_channels, 2 * res_channels, 3)
        if not unconditional:
            self.upsample_conv2d = upsampler_params(mel_upsample)
            self.mel_conv = ConvParams(80, 2 * res_channels, 1)  # 80 mel bands (`wavenet.py:70`)
        self.res_conv = WNParams((res_channels, res_channels, 1), res_channels, res_channels)
        self.skip_conv = WNParams((skip_channels, res_channels, 1), res_channels, skip_channels)


class _ResidualGroupParams(nn.Module):
    """Parameters of ``Residual_group`` (``models/wavenet.py:124-147``)."""

    def __init__(self, res_channels, skip_channels, num_res_layers, embed_in, embed_mid, embed_out,
                 unconditional, mel_upsample):
        super().__init__()
        self.fc_t1 = LinearParams(embed_in, embed_mid)
        self.fc_t2 = LinearParams(embed_mid, embed_out)
        self.residual_blocks = nn.ModuleList(
            _ResidualBlockParams(res_channels, skip_channels, embed_out, unconditional, mel_upsample)
            for _ in range(num_res_layers))


class WaveNet(EngineModule):
    def __init__(self, in_channels=1, res_channels=256, skip_channels=128, out_channels=1,
                 num_res_layers=30, dilation_cycle=10,
                 diffusion_step_embed_dim_in=128,
                 diffusion_step_embed_dim_mid=512,
                 diffusion_step_embed_dim_out=512,
                 unconditional=False,
                 mel_upsample=[16, 16],
                 **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.res_channels = res_channels
        self.skip_channels = skip_channels
        self.num_res_layers = num_res_layers
        self.dilation_cycle = dilation_cycle
        self.unconditional = unconditional
        self.mel_upsample = list(mel_upsample)
        self.embed_dims = (diffusion_step_embed_dim_in, diffusion_step_embed_dim_mid, diffusion_step_embed_dim_out)

        self.init_conv = nn.ModuleList([ConvParams(in_channels, res_channels, 1)])
        self.residual_layer = _ResidualGroupParams(res_channels, skip_channels, num_res_layers,
                                                   *self.embed_dims, unconditional, self.mel_upsample)
        # index 1 is the parameter-free ReLU of the reference's nn.Sequential (`wavenet.py:198-200`)
        self.final_conv = nn.ModuleList([ConvParams(skip_channels, skip_channels, 1), nn.Identity(),
                                         ZeroConvParams(skip_channels, out_channels)])

    def _desc(self):
        d = _lib.ModelDesc()
        d.kind = _lib.DWS_KIND_WAVENET
        d.in_channels, d.out_channels = self.in_channels, self.out_channels
        (d.diffusion_step_embed_dim_in, d.diffusion_step_embed_dim_mid,
         d.diffusion_step_embed_dim_out) = self.embed_dims
        d.unconditional = 1 if self.unconditional else 0
        d.mel_upsample[0], d.mel_upsample[1] = self.mel_upsample
        d.mel_bands = 80
        d.res_channels, d.skip_channels = self.res_channels, self.skip_channels
        d.num_res_layers, d.dilation_cycle = self.num_res_layers, self.dilation_cycle
        return d

    def __repr__(self):
        return f"wavenet_h{self.res_channels}_d{self.num_res_layers}_{'uncond' if self.unconditional else 'cond'}"

    @classmethod
    def name(cls, cfg):
        # the reference's classmethod reads an undefined `model_cfg` (`wavenet.py:216-220`); this is the intent
        return "wnet_h{}_d{}".format(cfg["res_channels"], cfg["num_res_layers"])
