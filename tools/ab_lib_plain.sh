#!/bin/bash
# Same-box A/B of two BUILDS of libdws.so WITHOUT the profiler (wall-clock ms per step only; rocprofv3 adds per-launch
# overhead that hides launch-count changes):   tools/ab_lib_plain.sh <rounds> <bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
export DWS_BENCH_NO_DP_OVERHEAD=1
for i in $(seq 1 $N); do
  for which in prev new; do
    if [ $which = prev ]; then export DWS_LIB=$R/tools/ab/libdws_prev.so; else unset DWS_LIB; fi
    python $R/bench.py "$@" --no-cpu-baseline --no-extra --no-full-loop --no-roofline > /tmp/ab_plain.log 2>&1
    echo "== $which: $(grep '^{' /tmp/ab_plain.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("final_loss"))')"
  done
done
