import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import cases
dev = torch.device("cuda:0")
cfg0, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_c64"]
cfg = dict(cfg0); cfg["num_res_layers"] = 2; cfg["dilation_cycle"] = 1
L2 = 256
net = cases.build_ours(cfg, wseed + 9).to(dev)
audio, steps = cases.wavenet_inputs(1, L2, 1, iseed + L2)
out = {}
with torch.no_grad():
    for prec in ("f32", "bf16x6"):
        net.set_option("precision", prec)
        e = net((audio.to(dev), steps.to(dev)))
        out[prec] = net.read_tap("x", (1, 64, L2)).clone()[0].cpu()
a, b = out["f32"], out["bf16x6"]
torch.set_printoptions(precision=5, linewidth=200)
for r in (0, 1, 4):
    print("row", r, "f32 ", a[r, 44:64])
    print("row", r, "bx6 ", b[r, 44:64])
    print("row", r, "diff", (b - a)[r, 44:64])
