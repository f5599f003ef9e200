#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "base:" "adlate:-DC6_AD_LATE" "noil:-DC6_NO_INTERLEAVE"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_sashimi_chain6="$flags"
  python diffwave-sashimi_amd/build.py > /dev/null 2>&1
  echo "== $name"
  python - <<'PY'
import sys, torch, time
sys.path.insert(0, '.')
import bench, ctypes
from diffwave_sashimi_amd import _lib
lib = _lib.load()
for name in ("unet_d64_n6_T200", "unet_d32_n6_T50_cond"):
    cfg = bench.CONFIGS[name]
    dev = torch.device("cuda")
    net = bench.build_model(cfg, dev)
    B, L = cfg["B"], cfg["L"]
    x = torch.randn(B, 1, L, device=dev); st = torch.full((B, 1), 7.0, device=dev)
    mel = (torch.rand(B, 80, cfg["Tmel"], device=dev) * 13.5 - 11.5) if "Tmel" in cfg else None
    net.set_option("precision", "bf16x6")
    with torch.no_grad():
        for _ in range(3): net((x, st), mel_spec=mel) if mel is not None else net((x, st))
        _lib.check(lib.dws_profile_enable(b"s4_tail_mfma_chain"))
        for _ in range(5): net((x, st), mel_spec=mel) if mel is not None else net((x, st))
        torch.cuda.synchronize()
        n = ctypes.c_int64(); tot = ctypes.c_double()
        _lib.check(lib.dws_profile_query(ctypes.byref(n), ctypes.byref(tot))); lib.dws_profile_disable()
    print(name, "chain6 tails: %d launches, avg %.1f us" % (n.value, tot.value / max(n.value, 1) * 1e3))
PY
done
