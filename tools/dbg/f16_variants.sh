#!/bin/bash
# usage: f16_variants.sh "name:flags" ...   builds wavenet_bx6.hip with the flags, runs the f16x3 tests once, times f16x3 (and bf16x6)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_wavenet_bx6="$flags"
  python diffwave-sashimi_amd/build.py > /tmp/build_$name.log 2>&1 || { echo "build failed $name"; tail -5 /tmp/build_$name.log; continue; }
  echo "== $name ($flags)"
  python -m pytest tests/test_f16x3_gpu.py -q -x 2>&1 | grep -E "passed|failed" | cut -c1-200 | head -3
  python tools/wn_layer_times.py --precision f16x3 --reps 3 2>&1 | tail -1
  [ -n "$BX6_TOO" ] && python tools/wn_layer_times.py --precision bf16x6 --reps 3 2>&1 | tail -1
done
