import sys, os, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import cases
from diffwave_sashimi_amd import _lib
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
gpu = torch.device("cuda")
lib = _lib.load()
cfg, B, L, wseed, _, _ = cases.WAVENET_CASES["wn_c64"]
net = cases.build_ours(cfg, wseed).to(gpu)
T = 3
dh = calc_diffusion_hyperparams(T, 1e-4, 0.05)
tabs = [np.ascontiguousarray(dh[k].numpy()) for k in ("Alpha", "Alpha_bar")] + [np.zeros(T, dtype=np.float32)]   # sigma = 0
ptabs = [t.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) for t in tabs]
al, ab = dh["Alpha"], dh["Alpha_bar"]
x0 = torch.randn(B, 1, L)
for t in range(T):
    x = x0.to(gpu).clone()
    net._sync_params(L); net._prepare(B, L)
    _lib.check(lib.dws_sampler_steps(net._handle, x.data_ptr(), *ptabs, T, t, 1, 99, 0, _lib.current_stream()))
    torch.cuda.synchronize()
    eps = net.read_tap("sampler_eps", (B, 1, L)).cpu().numpy()
    got = x.cpu().numpy()
    c1 = np.float32((1 - al[t]) / torch.sqrt(1 - ab[t])); c2 = np.float32(torch.sqrt(al[t]))
    xn = x0.numpy()
    p = c1 * eps
    want = (xn - p) / c2
    d = got != want
    print("t", t, "c1", float(c1), "c2", float(c2), "n diff", int(d.sum()), "of", d.size, flush=True)
    idx = np.argwhere(d)[:4]
    for i in idx:
        i = tuple(i)
        exact = (np.float64(xn[i]) - np.float64(p[i])) / np.float64(c2)
        fused = (np.float64(xn[i]) - np.float64(c1) * np.float64(eps[i])) / np.float64(c2)
        print("   x %.9g eps %.9g  gpu %.9g numpy %.9g  | exact-after-rounded-product %.12g  fully-exact %.12g" % (xn[i], eps[i], got[i], want[i], exact, fused))
