#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "grp:" "nogrp:-DWGRAD_NO_XCD_GROUP"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_wavenet_backward_mfma="$flags"
  python diffwave-sashimi_amd/build.py > /dev/null 2>&1
  echo "== $name"
  for P in f32 bf16x6; do
   DWS_BENCH_NO_DP_OVERHEAD=1 timeout 900 python bench.py --config unet_d128_n6_T200 --mode train --precision $P --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C5 train $P', round(d['ms_per_step'],2), d['final_loss'])"
  done
done
unset DWS_HIPCC_FLAGS_wavenet_backward_mfma; python diffwave-sashimi_amd/build.py > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_sashimi_training_gpu.py -x -q -k "bf16x6 or d32" 2>&1 | tail -2
bash tools/dbg/prof_tr6.sh | grep -E "wgrad" | cut -c1-140 | head -3
