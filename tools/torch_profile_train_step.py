import sys, torch
sys.path.insert(0, '.')
import bench
from torch.profiler import profile, ProfilerActivity
cfg = bench.CONFIGS["unet_d128_n6_T200"]
dev = torch.device("cuda:0")
import torch.nn as nn
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
from diffwave_sashimi_amd.training import training_loss
net = bench.build_model(cfg, dev).train()
opt = torch.optim.Adam(net.parameters(), lr=2e-4)
dh = calc_diffusion_hyperparams(**cfg["diffusion"])
g = torch.Generator().manual_seed(1)
B, L = 8, cfg["L"]
audio = ((torch.rand(B, 1, L, generator=g) * 2 - 1) * 0.3).to(dev)
loss_fn = nn.MSELoss()
def step():
    opt.zero_grad(set_to_none=True)
    loss = training_loss(net, loss_fn, audio, dh, generator=g)
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:40]:
    print(f"{e.key[:70]:70s} count={e.count:5d} cpu={e.cpu_time_total/1e3:8.2f}ms")
