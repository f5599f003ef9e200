#!/bin/bash
# same-box A/B of the operand-prefetch changes (tail kernel, tapconv, wgrad_dma)
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import json,sys; print(json.loads(sys.stdin.read())[\"ms_per_step\"])"; }
rebuild() { touch diffwave-sashimi_amd/csrc/$1.hip; python diffwave-sashimi_amd/build.py > /dev/null; }
for fl in "" "-DDWS_TAIL_NO_BPREFETCH"; do export DWS_HIPCC_FLAGS_sashimi_mfma="$fl"; rebuild sashimi_mfma
  echo "tail [$fl]: C3 $(run --config unet_d64_n6_T200 --steps 20 --warmup 3) $(run --config unet_d64_n6_T200 --steps 20 --warmup 3)  C4 $(run --config unet_d32_n6_T50_cond --steps 20 --warmup 3) $(run --config unet_d32_n6_T50_cond --steps 20 --warmup 3) d128 $(run --config unet_d128_n6_T200 --steps 10 --warmup 2)"; done
unset DWS_HIPCC_FLAGS_sashimi_mfma; rebuild sashimi_mfma
for fl in "" "-DDWS_TC_NO_BPREFETCH" "-DDWS_WG_NO_PREFETCH" "-DDWS_TC_NO_BPREFETCH -DDWS_WG_NO_PREFETCH" ""; do export DWS_HIPCC_FLAGS_wavenet_backward_mfma="$fl"; rebuild wavenet_backward_mfma
  echo "train [$fl]: C5 $(run --config unet_d128_n6_T200 --mode train --steps 4 --warmup 2) $(run --config unet_d128_n6_T200 --mode train --steps 4 --warmup 2)  WN $(run --config wnet_h256_d36_T200 --mode train --steps 6 --warmup 2)"; done
unset DWS_HIPCC_FLAGS_wavenet_backward_mfma; rebuild wavenet_backward_mfma
