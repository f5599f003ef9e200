"""Training-step pieces around the network (``train.py:198-222``): q-sample and the epsilon loss.  ``net`` is
any differentiable module of the reference surface; with the engine modules the call goes through
``models.engine._EngineTrainFn`` (``dws_model_forward_train`` / ``dws_model_backward``), which keeps the
activations of exactly one forward per module -- pair every forward with its backward."""
import torch


class _PinnedStager:
    """Host-to-device copies that do not block the host: the CPU tensor goes through a page-locked staging buffer
    (one per call site) and an asynchronous copy; an event guards the buffer's reuse.  A pageable ``.to(device)``
    makes the host wait until the GPU has drained everything queued before it -- once per training step that
    serialises the step's Python overhead (state-dict walk, ~2000 launches' worth of enqueueing) with the GPU."""

    def __init__(self):
        self._slots = {}

    def __call__(self, t, device, slot):
        device = torch.device(device)
        if device.type != "cuda" or t.device.type != "cpu":
            return t.to(device)
        ent = self._slots.get(slot)
        if ent is None or ent[0].shape != t.shape or ent[0].dtype != t.dtype:
            ent = [torch.empty(t.shape, dtype=t.dtype).pin_memory(), None]
            self._slots[slot] = ent
        if ent[1] is not None:
            ent[1].synchronize()          # the previous copy out of this buffer has run (a step ago)
        ent[0].copy_(t)
        out = ent[0].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ent[1] = ev
        return out


_stage = _PinnedStager()
_alpha_bar_on = {}   # (id of the CPU tensor, device) -> device copy of Alpha_bar


def _alpha_bar(Alpha_bar, device):
    device = torch.device(device)
    if Alpha_bar.device == device:
        return Alpha_bar
    key = (id(Alpha_bar), Alpha_bar._version, str(device))
    hit = _alpha_bar_on.get(key)
    if hit is None:
        _alpha_bar_on.clear()
        hit = _alpha_bar_on[key] = (Alpha_bar, Alpha_bar.to(device))   # keeps the CPU tensor alive: the id stays unique
    return hit[1]


def q_sample(audio, diffusion_steps, Alpha_bar, z):
    """x_t ~ q(x_t | x_0): ``sqrt(abar_t) x_0 + sqrt(1 - abar_t) z`` (``train.py:220``)."""
    ab = _alpha_bar(Alpha_bar, audio.device)[diffusion_steps]
    return torch.sqrt(ab) * audio + torch.sqrt(1 - ab) * z


def training_loss(net, loss_fn, audio, diffusion_hyperparams, mel_spec=None, generator=None):
    """``train.py:198-222``: ``t ~ U{0..T-1}``, ``z ~ N(0, I)``, ``loss_fn(net((x_t, t), mel), z)``.
    Steps and noise are drawn on the CPU generator exactly like the reference (then moved, through pinned staging
    buffers so the host does not stall), so a seeded call consumes the RNG stream in the same order."""
    T, Alpha_bar = diffusion_hyperparams["T"], diffusion_hyperparams["Alpha_bar"]
    B, C, L = audio.shape
    diffusion_steps = _stage(torch.randint(T, size=(B, 1, 1), generator=generator), audio.device, "steps")
    z = _stage(torch.normal(0, 1, size=audio.shape, generator=generator), audio.device, "z")
    x_t = q_sample(audio, diffusion_steps, Alpha_bar, z)
    epsilon_theta = net((x_t, diffusion_steps.view(B, 1)), mel_spec=mel_spec)
    return loss_fn(epsilon_theta, z)
