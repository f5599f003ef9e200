import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import cases
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
gpu = torch.device("cuda")
cfg, B, L, wseed, _, _ = cases.WAVENET_CASES["wn_c64"]
net = cases.build_ours(cfg, wseed).to(gpu)
NL, C = cfg["num_res_layers"], cfg["res_channels"]
T = 3
dh = calc_diffusion_hyperparams(T, 1e-4, 0.05)
x_T = torch.randn(B, 1, L)
sampling(net, (B, 1, L), dh, x_T=x_T, noise=torch.zeros(T, B, 1, L), use_graph=False)
tab = net.read_tap("tab_part_t", (T, NL * C)).cpu().numpy()
arow = 4 * 2 * C
tab_abt = net.read_tap("tab_abt", (NL, T, arow)).cpu().numpy()
for t in range(T):
    with torch.no_grad():
        net((x_T.to(gpu), torch.full((B, 1), float(t), device=gpu)))
    pt = net.read_tap("part_t", (B, NL * C)).cpu().numpy()
    abt = net.read_tap("abt", (NL, B, arow)).cpu().numpy()
    print("t", t, "part_t rows equal to table row:", [bool((pt[b] == tab[t]).all()) for b in range(B)],
          "max diff", float(np.abs(pt - tab[t]).max()), "| abt equal:", [bool((abt[:, b] == tab_abt[:, t]).all()) for b in range(B)],
          "max diff", float(np.abs(abt - tab_abt[:, t:t + 1]).max()), flush=True)
