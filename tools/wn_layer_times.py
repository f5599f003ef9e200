#!/usr/bin/env python
"""Per-dilation launch time of the fused WaveNet layer kernel (HIP events on the launch stream, eager launches).
usage (GPU box): python tools/wn_layer_times.py [--config wnet_h256_d36_T200] [--batch B] [--algo winograd|direct]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffwave_sashimi_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="wnet_h256_d36_T200")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--algo", default="winograd")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16x6", "f16x3"])
    args = ap.parse_args()
    cfg = dict(bench.CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    lib = _lib.load()
    dev = torch.device("cuda:0")
    net = bench.build_model(cfg, dev)
    net.set_option("conv_algo", args.algo)
    net.set_option("precision", args.precision)
    B, L = cfg["B"], cfg["L"]
    x = torch.randn(B, 1, L, device=dev)
    t = torch.full((B, 1), 17.0, device=dev)
    m = cfg["model"]
    NL, cyc = m["num_res_layers"], m["dilation_cycle"]
    with torch.no_grad():
        net((x, t))
        _lib.check(lib.dws_profile_enable(b"wn_layer"))
        for _ in range(args.reps):
            net((x, t))
        torch.cuda.synchronize()
        n = ctypes.c_int64()
        buf = (ctypes.c_double * (NL * args.reps))()
        _lib.check(lib.dws_profile_query_each(buf, NL * args.reps, ctypes.byref(n)))
        lib.dws_profile_disable()
    assert n.value == NL * args.reps, (n.value, NL, args.reps)
    per = {}
    for i in range(n.value):
        per.setdefault(1 << ((i % NL) % cyc), []).append(buf[i])
    tot = 0.0
    for d in sorted(per):
        v = sorted(per[d])
        print("d=%5d  n=%3d  median %.4f ms  min %.4f  max %.4f" % (d, len(v), v[len(v) // 2], v[0], v[-1]))
        tot += sum(v) / len(v) * (NL // cyc)
    print("sum over one forward (%d layers): %.3f ms (%s, %s)" % (NL, tot, args.algo, args.precision))


if __name__ == "__main__":
    main()
