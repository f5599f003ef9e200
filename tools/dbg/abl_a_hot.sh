#!/bin/bash
# GEMM1 of the split layer kernels with the weight fragments served from L1 (BX6_ABL_A_HOT: wrong results, timing only)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "stream:" "a_hot:-DBX6_ABL_A_HOT"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_wavenet_bx6="$flags"
  python diffwave-sashimi_amd/build.py > /dev/null 2>&1
  for P in f16x3 bf16x6; do
    echo "== $name $P"
    DWS_BX6_TRACE=1 python tools/wn_layer_times.py --precision $P --reps 1 2>&1 | grep "trace\] d=256 " | cut -c1-220
    python tools/wn_layer_times.py --precision $P --reps 3 2>&1 | tail -1
  done
done
unset DWS_HIPCC_FLAGS_wavenet_bx6; python diffwave-sashimi_amd/build.py > /dev/null 2>&1
