"""Import harness for the reference checkout (BUILD CONTAINER ONLY).

Used solely by ``make_golden.py`` to produce the committed ``*.npz`` vectors.
``/root/reference`` does not exist on the GPU box and nothing at test/bench
time imports this file.  It stubs the reference's absent third-party imports
in-process and restores the extension's symmetric
Cauchy semantics in the pure-torch fallback (semantic trap 1)."""
import os
import sys
import types

import torch

REF = os.environ.get("DWS_REFERENCE", "/root/reference")


def load():
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference checkout not found at {REF}")
    if "models.s4" in sys.modules:
        return sys.modules["models"], sys.modules["generate"], sys.modules["utils"], sys.modules["models.s4"]
    oe = types.ModuleType("opt_einsum")  # `models/s4.py:13-16`
    oe.contract = lambda eq, *ops, **kw: torch.einsum(eq, *ops)
    oe.contract_expression = lambda eq, *shapes, **kw: (lambda *ops: torch.einsum(eq, *ops))
    sys.modules["opt_einsum"] = oe
    pl = types.ModuleType("pytorch_lightning")  # `models/s4.py:11`
    plu = types.ModuleType("pytorch_lightning.utilities")
    plu.rank_zero_only = lambda f: f
    pl.utilities = plu
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.utilities"] = plu
    hy = types.ModuleType("hydra")  # `generate.py:12,203`
    hy.main = lambda **k: (lambda f: f)
    sys.modules["hydra"] = hy
    oc = types.ModuleType("omegaconf")
    oc.DictConfig = dict
    oc.OmegaConf = object
    sys.modules["omegaconf"] = oc
    torch.Tensor.cuda = lambda self, *a, **k: self  # `models/utils.py:24`, `generate.py:47-54`, `utils.py:150`
    sys.path.insert(0, REF)
    import models.s4 as s4
    _naive = s4.cauchy_naive  # `models/s4.py:109-116` lacks the conjugate half
    s4.cauchy_naive = lambda v, z, w: _naive(s4._conj(v), z, s4._conj(w))
    import models
    import generate
    import utils
    return models, generate, utils, s4
