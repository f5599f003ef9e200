"""One small SaShiMi forward through the fused FFT kernels against the oracle, with progress marks (debug aid, GPU box)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import cases
from oracle import sashimi as osa
gpu = torch.device("cuda")
for L in (1024, 4096, 16000):
    cfg = cases.ss_cfg(d_model=32, n_layers=1, L=L, diffusion_step_embed_dim_mid=64)
    net = cases.build_ours(cfg, 5).to(gpu)
    x = torch.randn(2, 1, L); st = torch.tensor([[3.0], [7.0]])
    print("L", L, "forward ...", flush=True)
    t0 = time.time()
    with torch.no_grad():
        out = net((x.to(gpu), st.to(gpu))).cpu()
    torch.cuda.synchronize()
    print("   done in %.2fs" % (time.time() - t0), flush=True)
    with torch.no_grad():
        ref = osa.sashimi_forward({k: v.cpu() for k, v in net.state_dict().items()}, cfg, x, st)
    print("   rel err", float((out - ref).abs().max() / ref.abs().max()), flush=True)
