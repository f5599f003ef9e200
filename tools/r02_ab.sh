#!/bin/bash
# Same-box A/B of build variants (box-to-box variance on this pool is +-5..8 %):
#   tools/r02_ab.sh <bench config> <file stem> name1:"flags" name2:"flags" ...
# builds csrc/<stem>.hip with each flag set in turn and prints the bench step time and the per-kernel rocprof averages.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
CFG=$1; STEM=$2; shift 2
cd /tmp && export TMPDIR=/tmp
DEF=$(python -c "import sys; sys.path.insert(0,'$R/diffwave-sashimi_amd'); import build; print(' '.join(build.FILE_FLAGS.get('$STEM', [])))")
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_$STEM="$DEF $flags"
  touch $R/diffwave-sashimi_amd/csrc/$STEM.hip; python $R/diffwave-sashimi_amd/build.py > /dev/null
  W=/tmp/prof_$name; rm -rf $W; mkdir -p $W
  rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline > $W/stats.log 2>&1
  echo "== $name ($flags): $(grep '^{' $W/stats.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | head -8 | cut -c1-150
  rm -rf $W
done
unset DWS_HIPCC_FLAGS_$STEM
touch $R/diffwave-sashimi_amd/csrc/$STEM.hip; python $R/diffwave-sashimi_amd/build.py > /dev/null
