"""Per-shape times of the symmetric Cauchy kernels at the shapes one config-5 training step launches them with
(`sashimi_model.hip: build_kernel / backward`: rows = 6 H with w broadcast over H, N = 32 conjugate pairs, L/2+1 bins)
-- through the public C-ABI (`dws_cauchy_sym_fwd/bwd`, w materialised per row), HIP events on the launch stream.
usage: python tools/cauchy_times.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffwave-sashimi_amd", "extensions"))
import cauchy_mult as cm  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def crand(*shape):
    return torch.view_as_complex(torch.randn(*shape, 2, device=dev, generator=g).contiguous())


for H, L in ((128, 16000), (256, 4000), (512, 1000), (64, 16000), (32, 16000)):
    B, N, Lh = 6 * H, 32, L // 2 + 1
    v, w = crand(B, N), crand(B, N)
    w = torch.complex(-w.real.abs() - 0.1, w.imag * 30)
    z = torch.complex(torch.zeros(Lh, device=dev), torch.linspace(-300, 300, Lh, device=dev))
    dout = crand(B, Lh)
    for name, fn in (("fwd", lambda: cm.cauchy_mult_sym_fwd(v, z, w)), ("bwd", lambda: cm.cauchy_mult_sym_bwd(v, z, w, dout))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        pairs = B * N * Lh
        print(f"H={H:4d} L={L:6d} rows={B:5d} N={N} bins={Lh:5d}  {name}: {us:8.1f} us   {pairs / us * 1e-3:7.2f} G pairs/s")
