"""Per-phase tick budget of the fused S4 convolution kernel (DWS_FFT_TRACE=1: s_memtime stamps of every wave for the first
rows a workgroup walks): one eager forward of a bench config; the summary lines come from the library on stderr.
    python tools/fft_trace.py [config] 2> trace.txt"""
import os
import sys
os.environ["DWS_FFT_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "unet_d64_n6_T200"
cfg = bench.CONFIGS[name]
dev = torch.device("cuda")
net = bench.build_model(cfg, dev)
x = torch.randn(cfg["B"], 1, cfg["L"], device=dev)
st = torch.full((cfg["B"], 1), 7.0, device=dev)
with torch.no_grad():
    for _ in range(2):          # the second forward runs warm
        sys.stderr.write("---- forward\n")
        net((x, st))
        torch.cuda.synchronize()
