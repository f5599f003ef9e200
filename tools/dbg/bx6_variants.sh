#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "nop1:-DBX6_DBG_STORE_NOP=0" "nop2:-DBX6_DBG_STORE_NOP=1" "nop4:-DBX6_DBG_STORE_NOP=3" "nop8:-DBX6_DBG_STORE_NOP=7" "nop1_red:-DBX6_DBG_STORE_NOP=0 -DBX6_DBG_REDUNDANT_LOAD" "nop2_red:-DBX6_DBG_STORE_NOP=1 -DBX6_DBG_REDUNDANT_LOAD"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_wavenet_bx6="$flags"
  python diffwave-sashimi_amd/build.py > /dev/null 2>&1
  echo "== $name"; for k in 1 2; do python -m pytest tests/test_bf16x6_gpu.py -q -k "h256 or c128 or c64" 2>&1 | grep -E "passed|failed" | head -5; done
done
