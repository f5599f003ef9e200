"""MI355X-native DiffWave denoising-loop engine (hand-written HIP for gfx950
behind the reference's model/operator surface).

Public surface (mirrors albertfgu/diffwave-sashimi):
  * ``models.construct_model(model_cfg)``          -- ``models/__init__.py:4-12``
  * ``models.wavenet.WaveNet`` / ``models.sashimi.Sashimi`` -- drop-ins for the
    reference modules (same constructor kwargs, forward signature, state_dict keys)
  * ``extensions.cauchy.cauchy_mult``              -- ``extensions/cauchy/cauchy.py:46-63``
  * ``sampling.sampling`` / ``sampling.calc_diffusion_hyperparams`` -- ``generate.py:23-55``, ``utils.py:121-151``

All compute goes through ``libdws.so`` (C ABI in ``include/dws.h``); there is
no CPU or eager-PyTorch fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["models", "extensions", "sampling"]
