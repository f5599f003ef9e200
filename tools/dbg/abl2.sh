#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "base:" "no_dma_wait:-DBX6_ABL_NO_DMA_WAIT" "no_stage:-DBX6_ABL_NO_STAGE" "no_barrier:-DBX6_ABL_NO_BARRIER -DBX6_ABL_NO_DMA_WAIT" "no_transform:-DBX6_ABL_NO_TRANSFORM" "all:-DBX6_ABL_NO_TRANSFORM -DBX6_ABL_NO_STAGE -DBX6_ABL_NO_BARRIER -DBX6_ABL_NO_DMA_WAIT -DBX6_ABL_A_HOT"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_wavenet_bx6="$flags"
  python diffwave-sashimi_amd/build.py > /dev/null 2>&1
  for P in f16x3; do
    echo "== $name $P"
    DWS_BX6_TRACE=1 timeout 120 python tools/wn_layer_times.py --precision $P --reps 1 2>&1 | grep "trace\] d=256 " | head -2 | cut -c1-220
  done
done
unset DWS_HIPCC_FLAGS_wavenet_bx6; python diffwave-sashimi_amd/build.py > /dev/null 2>&1
