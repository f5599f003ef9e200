#!/bin/bash
# tail kernel tile shapes, same box: DWS_TAIL_CFG=0 (current shapes) vs 1 (round-1 shapes) x config; per-kernel rocprof averages.
# (The sweep that chose the shapes also varied the template arguments and the k-group loop unrolling by editing
# launch_s4_tail_mfma; its numbers are in profiles/r02_tail_shapes.txt.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in unet_d64_n6_T200 unet_d32_n6_T50_cond; do for alt in 0 1; do
  W=/tmp/prof_t$alt; rm -rf $W; mkdir -p $W
  DWS_TAIL_CFG=$alt rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $W/log 2>&1
  echo "== $cfg DWS_TAIL_CFG=$alt: $(grep '^{' $W/log | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | grep s4_tail | cut -c1-150
  rm -rf $W
done; done
