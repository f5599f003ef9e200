import sys, math, torch, torch.nn as nn
sys.path.insert(0, '/root/repo')
from tests import cases
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
from diffwave_sashimi_amd.training import training_loss
dev = torch.device('cuda', 0)
which = sys.argv[1]
if which == 'wavenet':
    cfg = cases.wn_cfg(res_channels=64, skip_channels=64, num_res_layers=8, dilation_cycle=8)
    L = 2048
else:
    cfg = cases.ss_cfg(d_model=32, n_layers=2, L=2048, diffusion_step_embed_dim_mid=128)
    L = 2048
torch.manual_seed(0)
from diffwave_sashimi_amd.models import construct_model
net = construct_model(dict(cfg)).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=2e-3 if which == 'wavenet' else 1e-3)
dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
g = torch.Generator().manual_seed(1)
t = torch.arange(L) / 16000.0
def batch(B=8):
    f = 200 + 600 * torch.rand(B, 1, generator=g)
    ph = 2 * math.pi * torch.rand(B, 1, generator=g)
    return (0.5 * torch.sin(2 * math.pi * f * t[None] + ph)).unsqueeze(1).to(dev)
hist = []
for it in range(int(sys.argv[2])):
    opt.zero_grad()
    loss = training_loss(net, nn.MSELoss(), batch(), dh, generator=g)
    loss.backward()
    opt.step()
    hist.append(float(loss))
    if it % 50 == 0 or it == int(sys.argv[2]) - 1:
        print(it, round(sum(hist[-20:]) / len(hist[-20:]), 4), flush=True)
net.eval()
x = sampling(net, (4, 1, L), dh, seed=3)
print("sample stats: std", float(x.std()), "absmax", float(x.abs().max()), "finite", bool(torch.isfinite(x).all()))
# spectral peak of generated samples (a trained model should produce near-sinusoids in 200..800 Hz)
X = torch.fft.rfft(x[:, 0].cpu(), dim=-1).abs()
pk = X.argmax(dim=-1) * 16000.0 / L
print("dominant frequencies (Hz):", [round(float(p)) for p in pk], "peak/total energy", [round(float(X[i].max()**2 / (X[i]**2).sum()), 3) for i in range(4)])
