#!/bin/bash
# Same-box A/B of an environment switch of ONE build, wall clock only (no profiler):
#   tools/ab_env_plain.sh <rounds> <VAR> <bench args...>     "old" = VAR=1, "new" = VAR unset
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; VAR=$2; shift 2
export DWS_BENCH_NO_DP_OVERHEAD=1
for i in $(seq 1 $N); do
  for which in old new; do
    if [ $which = old ]; then export $VAR=1; else unset $VAR; fi
    python $R/bench.py "$@" --no-cpu-baseline --no-extra --no-full-loop --no-roofline > /tmp/ab_plain.log 2>&1
    echo "== $which ($VAR=${!VAR:-unset}): $(grep '^{' /tmp/ab_plain.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("final_loss"))')"
  done
done
