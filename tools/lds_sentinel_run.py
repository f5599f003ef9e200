"""Run LDS / register / global-memory sentinels (tools/ubench/lds_sentinel.hip) on a side stream beside config-5-shaped training
steps:   python tools/lds_sentinel_run.py [config] [batch] [precision]"""
import ctypes, os, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
from diffwave_sashimi_amd.training import training_loss
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
lib = ctypes.CDLL(os.path.join(R, "tools/ubench/liblds_sentinel.so"))
lib.sentinel_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
name = sys.argv[1] if len(sys.argv) > 1 else "unet_d128_n6_T200"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
cfg = dict(bench.CONFIGS[name]); dev = torch.device("cuda", 0)
net = bench.build_model(cfg, dev).train()
net.set_option("precision", prec)
opt = torch.optim.Adam(net.parameters(), lr=2e-4)
dh = calc_diffusion_hyperparams(**cfg["diffusion"])
audio = ((torch.rand(B, 1, cfg["L"]) * 2 - 1) * 0.3).to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = training_loss(net, nn.MSELoss(), audio, dh)
    loss.backward(); opt.step()
    return loss
os.environ["DWS_TRAIN_SERIAL_KERNELS"] = "1"
for _ in range(2): step()
torch.cuda.synchronize()
NL, NWG = 4000, 512
counters = torch.zeros(3 * NL, dtype=torch.int32, device=dev)
slab = torch.zeros(NWG * 64 * 4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(NL):
    lib.sentinel_launch(ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(counters.data_ptr()), ctypes.c_void_p(slab.data_ptr()), i, NWG, 4000)
t1 = time.perf_counter()
for _ in range(2): step()
torch.cuda.synchronize()
t2 = time.perf_counter()
c = counters.view(NL, 3).cpu()
bad = (c != 0).any(dim=1).nonzero().flatten().tolist()
print(f"{name} B={B} {prec}: {NL} sentinel launches enqueued in {1e3*(t1-t0):.1f} ms, 2 steps + sentinels done in {1e3*(t2-t1):.1f} ms")
print("launches with a corrupted LDS pattern:", int((c[:, 0] != 0).sum()), "| register:", int((c[:, 1] != 0).sum()), "| global slab:", int((c[:, 2] != 0).sum()))
print("first bad launches:", bad[:20], [c[i].tolist() for i in bad[:5]])
