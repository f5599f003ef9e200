#!/bin/bash
# Same-box sweep of one environment variable over values:  tools/sweep_env.sh <VAR> "<v1 v2 ...>" <kernel-substring> <bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2; KSUB=$3; shift 3
cd /tmp && export TMPDIR=/tmp
export DWS_BENCH_NO_DP_OVERHEAD=1
for rep in 1 2; do
for v in $VALS; do
  export $VAR=$v
  W=/tmp/sw_$v; rm -rf $W; mkdir -p $W
  rocprofv3 --kernel-trace --stats -d $W -o s -- python $R/bench.py "$@" --no-cpu-baseline --no-extra --no-full-loop --no-roofline > $W/log 2>&1
  echo "== $VAR=$v: $(grep '^{' $W/log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  python $R/tools/rocpd_summary.py stats $W/s_results.db | grep -i "$KSUB" | cut -c1-160
  rm -rf $W
done
done
