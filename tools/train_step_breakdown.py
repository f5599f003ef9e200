"""Where a training step spends its wall time (host-side parameter sync vs engine forward/backward vs optimizer):
    python tools/train_step_breakdown.py <bench config> <per-GPU batch>"""
import time, torch, torch.nn as nn, sys
sys.path.insert(0, '/root/repo')
import bench
from diffwave_sashimi_amd.models import engine
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
from diffwave_sashimi_amd.training import training_loss
cfg = dict(bench.CONFIGS[sys.argv[1]])
dev = torch.device("cuda", 0)
net = bench.build_model(cfg, dev).train()
opt = torch.optim.Adam(net.parameters(), lr=2e-4)
dh = calc_diffusion_hyperparams(**cfg["diffusion"])
B = int(sys.argv[2])
audio = ((torch.rand(B, 1, cfg["L"]) * 2 - 1) * 0.3).to(dev)
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
        T[name] = T.get(name, 0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
wrap(net, "_sync_params")
orig_bwd = engine._EngineTrainFn.backward
for it in range(4):
    if it == 1: T.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = training_loss(net, nn.MSELoss(), audio, dh)
    torch.cuda.synchronize(); tb = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); T["backward_total"] = T.get("backward_total", 0) + time.perf_counter() - tb
    ts = time.perf_counter(); opt.step(); torch.cuda.synchronize(); T["opt"] = T.get("opt", 0) + time.perf_counter() - ts
torch.cuda.synchronize()
print("step ms", (time.perf_counter() - t0) / 3 * 1e3, {k: round(v / 3 * 1e3, 2) for k, v in T.items()})
