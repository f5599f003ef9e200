"""where does the bf16x6 layer differ from the f32 path? (debug)  usage: bx6_diff.py case L B"""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import cases
dev = torch.device("cuda:0")
name = sys.argv[1]; L2 = int(sys.argv[2]); B2 = int(sys.argv[3])
cfg0, B, L, wseed, iseed, _ = cases.WAVENET_CASES[name]
C, S = cfg0["res_channels"], cfg0["skip_channels"]
for NL in (1, 2, 3, 4, 5, 6, 8, 12, 13):
    cfg = dict(cfg0); cfg["num_res_layers"] = NL
    net = cases.build_ours(cfg, wseed + 9).to(dev)
    audio, steps = cases.wavenet_inputs(B2, L2, 1, iseed + L2)
    out = {}
    with torch.no_grad():
        for prec in ("f32", "bf16x6", "bf16x6b"):
            net.set_option("precision", prec[:6])
            e = net((audio.to(dev), steps.to(dev)))
            out[prec] = (e.clone(), net.read_tap("x", (B2, C, L2)).clone(), net.read_tap("skip", (B2, S, L2)).clone())
    line = "NL %d:" % NL
    for k, nm in ((1, "x"), (2, "skip")):
        a, b, b2 = out["f32"][k], out["bf16x6"][k], out["bf16x6b"][k]
        d = (a - b).abs() / a.abs().max()
        bad = (d > 1e-4).nonzero()
        line += " %s rel %.1e det %s nbad %d" % (nm, float(d.max()), bool(torch.equal(b, b2)), len(bad))
        if len(bad):
            line += " b%s rows%s cols%s" % (sorted(set(bad[:, 0].tolist()))[:4], sorted(set(bad[:, 1].tolist()))[:12], sorted(set(bad[:, 2].tolist()))[:24])
    print(line)
