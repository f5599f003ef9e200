"""Model registry with the reference's surface (``models/__init__.py:4-23``)."""
from .wavenet import WaveNet

try:
    from .sashimi import Sashimi
except ImportError:  # pragma: no cover - until the SaShiMi shim lands
    Sashimi = None


def _registry():
    return {"wavenet": WaveNet, "sashimi": Sashimi}


def construct_model(model_cfg):
    """``model._name_ in {wavenet, sashimi}`` -> module; pops and restores ``_name_``
    exactly like the reference so the caller's config object is unchanged."""
    name = model_cfg.pop("_name_")
    try:
        model_cls = _registry()[name]
        if model_cls is None:
            raise NotImplementedError(f"backbone '{name}' is not available in this build")
        model = model_cls(**model_cfg)
    finally:
        model_cfg["_name_"] = name  # restore
    return model


def model_identifier(model_cfg):
    name = model_cfg["_name_"] if isinstance(model_cfg, dict) else model_cfg._name_
    return _registry()[name].name(model_cfg)
