"""Per-phase tick budget of the S4 tail kernels (DWS_TAIL_TRACE=1: s_memtime stamps of every wave): one eager forward of a
bench config; the summary lines come from the library on stderr.   python tools/tail_trace.py [config [precision]] 2> trace.txt"""
import os
import sys
os.environ["DWS_TAIL_TRACE"] = "1"
os.environ["DWS_CHAIN_TRACE"] = "1"      # the register-chained tails (H <= 64) stamp their phases too
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "unet_d64_n6_T200"
cfg = bench.CONFIGS[name]
dev = torch.device("cuda")
net = bench.build_model(cfg, dev)
if len(sys.argv) > 2:
    net.set_option("precision", sys.argv[2])      # e.g. bf16x6: the split chain tails (sashimi_chain6.hip)
x = torch.randn(cfg["B"], 1, cfg["L"], device=dev)
st = torch.full((cfg["B"], 1), 7.0, device=dev)
with torch.no_grad():
    for _ in range(2):          # the second forward runs warm
        sys.stderr.write("---- forward\n")
        net((x, st))
        torch.cuda.synchronize()
