cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_sashimi_bf16x6_gpu.py -q -s 2>&1 | grep -E "^ss_|^\.ss|passed|failed|Error|assert " | cut -c1-400 | head -30
python - <<'PY'
import sys, torch, time
sys.path.insert(0, '.')
import bench, ctypes
from diffwave_sashimi_amd import _lib
lib = _lib.load()
for name in ("unet_d64_n6_T200", "unet_d128_n6_T200", "unet_d32_n6_T50_cond"):
    cfg = bench.CONFIGS[name]
    dev = torch.device("cuda")
    net = bench.build_model(cfg, dev)
    B, L = cfg["B"], cfg["L"]
    x = torch.randn(B, 1, L, device=dev); st = torch.full((B, 1), 7.0, device=dev)
    mel = (torch.rand(B, 80, cfg["Tmel"], device=dev) * 13.5 - 11.5) if "Tmel" in cfg else None
    for prec in ("f32", "bf16x6"):
        net.set_option("precision", prec)
        with torch.no_grad():
            for _ in range(3): net((x, st), mel_spec=mel) if mel is not None else net((x, st))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): net((x, st), mel_spec=mel) if mel is not None else net((x, st))
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
            _lib.check(lib.dws_profile_enable(b"s4_tail"))
            for _ in range(3): net((x, st), mel_spec=mel) if mel is not None else net((x, st))
            torch.cuda.synchronize()
            n = ctypes.c_int64(); tot = ctypes.c_double()
            _lib.check(lib.dws_profile_query(ctypes.byref(n), ctypes.byref(tot))); lib.dws_profile_disable()
        print(name, prec, "eager forward %.3f ms; all tails: %d launches/fwd, %.3f ms per forward" % (ms, n.value // 3, tot.value / 3))
PY
