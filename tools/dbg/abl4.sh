#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "base:" "bq_ahead:-DBX6_BQ_AHEAD" "bq_ahead_pf1:-DBX6_BQ_AHEAD -DBX6_PF2=1"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_wavenet_bx6="$flags"
  python diffwave-sashimi_amd/build.py > /dev/null 2>&1
  echo "== $name"
  python -m pytest tests/test_f16x3_gpu.py -q -x 2>&1 | tail -1
  DWS_BX6_TRACE=1 timeout 120 python tools/wn_layer_times.py --precision f16x3 --reps 1 2>&1 | grep "trace\] d=256 " | head -1 | cut -c1-220
  python tools/wn_layer_times.py --precision f16x3 --reps 3 2>&1 | tail -1
done
unset DWS_HIPCC_FLAGS_wavenet_bx6; python diffwave-sashimi_amd/build.py > /dev/null 2>&1
