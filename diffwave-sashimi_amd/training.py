"""Training-step pieces around the network (``train.py:198-222``): q-sample and the epsilon loss.  ``net`` is
any differentiable module of the reference surface; with the engine modules the call goes through
``models.engine._EngineTrainFn`` (``dws_model_forward_train`` / ``dws_model_backward``), which keeps the
activations of exactly one forward per module -- pair every forward with its backward."""
import torch


def q_sample(audio, diffusion_steps, Alpha_bar, z):
    """x_t ~ q(x_t | x_0): ``sqrt(abar_t) x_0 + sqrt(1 - abar_t) z`` (``train.py:220``)."""
    ab = Alpha_bar.to(audio.device)[diffusion_steps]
    return torch.sqrt(ab) * audio + torch.sqrt(1 - ab) * z


def training_loss(net, loss_fn, audio, diffusion_hyperparams, mel_spec=None, generator=None):
    """``train.py:198-222``: ``t ~ U{0..T-1}``, ``z ~ N(0, I)``, ``loss_fn(net((x_t, t), mel), z)``.
    Steps and noise are drawn on the CPU generator exactly like the reference (then moved), so a
    seeded call consumes the RNG stream in the same order."""
    T, Alpha_bar = diffusion_hyperparams["T"], diffusion_hyperparams["Alpha_bar"]
    B, C, L = audio.shape
    diffusion_steps = torch.randint(T, size=(B, 1, 1), generator=generator).to(audio.device)
    z = torch.normal(0, 1, size=audio.shape, generator=generator).to(audio.device)
    x_t = q_sample(audio, diffusion_steps, Alpha_bar, z)
    epsilon_theta = net((x_t, diffusion_steps.view(B, 1)), mel_spec=mel_spec)
    return loss_fn(epsilon_theta, z)
