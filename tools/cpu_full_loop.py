"""the full T-step reverse-diffusion loop of config 1 (wnet_h128_d30, B = 1, L = 16000) on the host CPU --
the oracle's network (reference-equivalent PyTorch-CPU graph) inside the reference's loop (`generate.py:47-54`) -- timed end to
end, beside its per-step extrapolation.   python tools/cpu_full_loop.py [threads] > profiles/r04_cpu_full_loop_c1.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from oracle import diffusion as odiff
from oracle import wavenet as own
from diffwave_sashimi_amd.models import construct_model

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.set_num_threads(threads)
cfg = bench.CONFIGS["wnet_h128_d30_T200"]
torch.manual_seed(0)
net = construct_model(dict(cfg["model"]))
sd = {k: v.detach() for k, v in net.state_dict().items()}
L, T = cfg["L"], cfg["diffusion"]["T"]
dh = odiff.calc_diffusion_hyperparams(**cfg["diffusion"])
fwd = lambda inp, mel_spec=None: own.wavenet_forward(sd, cfg["model"], inp[0], inp[1], mel_spec=mel_spec)
with torch.no_grad():
    x = torch.randn(1, 1, L)
    st = torch.full((1, 1), float(T - 1))
    fwd((x, st))                                   # warm-up
    t0 = time.perf_counter()
    for _ in range(3):
        fwd((x, st))
    per_step = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    out = odiff.sampling(fwd, (1, 1, L), dh)
    loop = time.perf_counter() - t0
print(json.dumps({"config": "wnet_h128_d30_T200", "B": 1, "L": L, "T": T, "threads": threads, "host_cpus": os.cpu_count(),
                  "cpu_model": bench._cpu_model(), "full_loop_s": loop, "samples_per_s": L / loop,
                  "forward_ms_per_step": per_step * 1e3, "extrapolated_loop_s": per_step * T,
                  "loop_over_extrapolation": loop / (per_step * T), "finite": bool(torch.isfinite(out).all())}))
