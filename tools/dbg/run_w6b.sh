cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import sys, torch, time
sys.path.insert(0, '.')
import bench, ctypes
from diffwave_sashimi_amd import _lib
lib = _lib.load()
for name in ("unet_d128_n6_T200", "unet_d64_n6_T200"):
    cfg = bench.CONFIGS[name]
    dev = torch.device("cuda")
    net = bench.build_model(cfg, dev)
    B, L = cfg["B"], cfg["L"]
    x = torch.randn(B, 1, L, device=dev); st = torch.full((B, 1), 7.0, device=dev)
    for prec in ("f32", "bf16x6"):
        net.set_option("precision", prec)
        with torch.no_grad():
            for _ in range(3): net((x, st))
            _lib.check(lib.dws_profile_enable(b"s4_tail"))
            for _ in range(3): net((x, st))
            torch.cuda.synchronize()
            n = ctypes.c_int64(); buf = (ctypes.c_double * 90)()
            _lib.check(lib.dws_profile_query_each(buf, 90, ctypes.byref(n))); lib.dws_profile_disable()
        per = [min(buf[r * 30 + i] for r in range(3)) * 1e3 for i in range(30)]
        print(name, prec, " ".join("%.0f" % v for v in per))
PY
