"""Where does the step-table sampler differ from the per-step loop?  (debug aid, GPU box)"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import cases
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling

gpu = torch.device("cuda")
for backbone in ("wavenet", "sashimi"):
    if backbone == "wavenet":
        cfg, B, L, wseed, _, _ = cases.WAVENET_CASES["wn_c64"]
    else:
        cfg, B, L, wseed = cases.ss_cfg(d_model=32, n_layers=2, L=1024, diffusion_step_embed_dim_mid=64), 3, 1024, 5
    net = cases.build_ours(cfg, wseed).to(gpu)
    for T in (1, 2, 5):
        dh = calc_diffusion_hyperparams(T, 1e-4, 0.05)
        g = torch.Generator().manual_seed(77)
        x_T, noise = torch.randn(B, 1, L, generator=g), torch.randn(T, B, 1, L, generator=g)
        al, ab, sg = (dh[k] for k in ("Alpha", "Alpha_bar", "Sigma"))
        for use_graph in (False, True):
            got = sampling(net, (B, 1, L), dh, x_T=x_T, noise=noise, use_graph=use_graph).cpu().numpy()
            x = x_T.numpy().copy()
            with torch.no_grad():
                for t in range(T - 1, -1, -1):
                    eps = net((torch.from_numpy(x).to(gpu), torch.full((B, 1), float(t), device=gpu))).cpu().numpy()
                    c1 = np.float32((1 - al[t]) / torch.sqrt(1 - ab[t]))
                    c2 = np.float32(torch.sqrt(al[t]))
                    x = (x - c1 * eps) / c2
                    if t > 0:
                        x = x + np.float32(sg[t]) * noise[t].numpy()
            d = np.abs(got - x)
            print(backbone, "T", T, "graph", use_graph, "max diff", d.max(), "n diff", int((d > 0).sum()), "of", d.size, flush=True)
    # the forward itself: twice the same call, and the same clip at another batch position
    xx = torch.randn(B, 1, L, device=gpu)
    st = torch.full((B, 1), 3.0, device=gpu)
    with torch.no_grad():
        a = net((xx, st)); b = net((xx, st))
        c = net((xx.flip(0).contiguous(), st)).flip(0)
    print(backbone, "forward repeat equal", bool(torch.equal(a, b)), "batch-position equal", bool(torch.equal(a, c)), flush=True)
