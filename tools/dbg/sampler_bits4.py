import sys, os, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import cases
from diffwave_sashimi_amd import _lib
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
gpu = torch.device("cuda")
lib = _lib.load()
for backbone in ("wavenet", "sashimi"):
    if backbone == "wavenet":
        cfg, B, L, wseed, _, _ = cases.WAVENET_CASES["wn_c64"]
    else:
        cfg, B, L, wseed = cases.ss_cfg(d_model=32, n_layers=2, L=1024, diffusion_step_embed_dim_mid=64), 3, 1024, 5
    net = cases.build_ours(cfg, wseed).to(gpu)
    T = 3
    dh = calc_diffusion_hyperparams(T, 1e-4, 0.05)
    tabs = [np.ascontiguousarray(dh[k].numpy()) for k in ("Alpha", "Alpha_bar", "Sigma")]
    ptabs = [t.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) for t in tabs]
    x0 = torch.randn(B, 1, L)
    for t in range(T):
        x = x0.to(gpu).clone()
        net._sync_params(L); net._prepare(B, L)
        # ONE eager reverse step at step index t (its Philox noise does not matter: only eps is read back)
        _lib.check(lib.dws_sampler_steps(net._handle, x.data_ptr(), *ptabs, T, t, 1, 99, 0, _lib.current_stream()))
        torch.cuda.synchronize()
        eps_s = net.read_tap("sampler_eps", (B, 1, L)).cpu()
        with torch.no_grad():
            eps_m = net((x0.to(gpu), torch.full((B, 1), float(t), device=gpu))).cpu()
        d = (eps_s - eps_m).abs()
        print(backbone, "t", t, "eps: sampler vs module call: max diff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel(), flush=True)
