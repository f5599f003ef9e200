import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import cases
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
gpu = torch.device("cuda")
cfg, B, L, wseed, _, _ = cases.WAVENET_CASES["wn_c64"]
net = cases.build_ours(cfg, wseed).to(gpu)
T = 2
dh = calc_diffusion_hyperparams(T, 1e-4, 0.05)
g = torch.Generator().manual_seed(77)
x_T, noise = torch.randn(B, 1, L, generator=g), torch.randn(T, B, 1, L, generator=g)
al, ab, sg = (dh[k] for k in ("Alpha", "Alpha_bar", "Sigma"))
print("dtypes", al.dtype, ab.dtype, sg.dtype, "sigma", [float(v) for v in sg], flush=True)
def loop(nz, tmap=lambda t: t):
    x = x_T.numpy().copy()
    with torch.no_grad():
        for t in range(T - 1, -1, -1):
            eps = net((torch.from_numpy(x).to(gpu), torch.full((B, 1), float(tmap(t)), device=gpu))).cpu().numpy()
            c1 = np.float32((1 - al[t]) / torch.sqrt(1 - ab[t])); c2 = np.float32(torch.sqrt(al[t]))
            x = (x - c1 * eps) / c2
            if t > 0:
                x = x + np.float32(sg[t]) * nz[t].numpy()
    return x
for name, nz in (("noise", noise), ("zero noise", torch.zeros_like(noise))):
    got = sampling(net, (B, 1, L), dh, x_T=x_T, noise=nz, use_graph=False).cpu().numpy()
    d = np.abs(got - loop(nz))
    print(name, "max diff", d.max(), "n diff", int((d > 0).sum()), flush=True)
# does the sampler's network see step t (as it should), or t-1 / t+1?
got = sampling(net, (B, 1, L), dh, x_T=x_T, noise=torch.zeros_like(noise), use_graph=False).cpu().numpy()
for shift in (-1, 0, 1):
    d = np.abs(got - loop(torch.zeros_like(noise), lambda t: t + shift))
    print("steps shifted by", shift, "max diff", d.max(), flush=True)
