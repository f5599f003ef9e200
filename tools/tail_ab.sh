#!/bin/bash
# per-kernel rocprof averages of the S4 tail kernels on C3 / C4 (one box): tools/tail_ab.sh [ENV=1 ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in unet_d64_n6_T200 unet_d32_n6_T50_cond; do
  rm -rf /tmp/pp; env "$@" rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $R/bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/pp.log 2>&1
  echo "== $cfg $* $(grep '^{' /tmp/pp.log | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  python $R/tools/rocpd_summary.py stats /tmp/pp/s_results.db | grep "s4_tail" | cut -c1-150
done
