#!/bin/bash
# Same-box A/B of two BUILDS of libdws.so (box-to-box variance on this pool is +-5..8 %):
#   tools/ab_lib.sh <kernel-substring> <bench args...>
# runs bench.py under rocprofv3 --kernel-trace --stats once with tools/ab/libdws_prev.so (DWS_LIB) and once with the
# in-tree library, prints ms/step and the matching kernels' average durations.
R=${GRAFT_REPO_ROOT:-/root/repo}
KSUB=$1; shift
cd /tmp && export TMPDIR=/tmp
for which in prev new prev new; do
  if [ $which = prev ]; then export DWS_LIB=$R/tools/ab/libdws_prev.so; else unset DWS_LIB; fi
  W=/tmp/ab_$which; rm -rf $W; mkdir -p $W
  rocprofv3 --kernel-trace --stats -d $W -o s -- python $R/bench.py "$@" --no-cpu-baseline --no-extra --no-full-loop > $W/log 2>&1
  echo "== $which: $(grep '^{' $W/log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  python $R/tools/rocpd_summary.py stats $W/s_results.db | grep -i "$KSUB" | cut -c1-160
  rm -rf $W
done
