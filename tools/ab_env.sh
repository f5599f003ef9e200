#!/bin/bash
# Same-box A/B of an environment switch of ONE build:  tools/ab_env.sh <VAR> <kernel-substring> <bench args...>
# runs bench.py under rocprofv3 --kernel-trace --stats with VAR=1 ("old") and without ("new"), twice each.
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; KSUB=$2; shift 2
cd /tmp && export TMPDIR=/tmp
export DWS_BENCH_NO_DP_OVERHEAD=1
for which in old new old new; do
  if [ $which = old ]; then export $VAR=1; else unset $VAR; fi
  W=/tmp/ab_$which; rm -rf $W; mkdir -p $W
  rocprofv3 --kernel-trace --stats -d $W -o s -- python $R/bench.py "$@" --no-cpu-baseline --no-extra --no-full-loop --no-roofline > $W/log 2>&1
  echo "== $which ($VAR=${!VAR:-unset}): $(grep '^{' $W/log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  python $R/tools/rocpd_summary.py stats $W/s_results.db | grep -i "$KSUB" | cut -c1-160
  rm -rf $W
done
