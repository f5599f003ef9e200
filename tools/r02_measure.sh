#!/bin/bash
# usage (GPU box): tools/r02_measure.sh <tag> <what...>   what in: c3 c4 c5train c2 wntrain d128
# Per config: the bench line (no profiler) -> gpurun_out/<tag>_bench_<what>.json, and rocprofv3 kernel stats ->
# gpurun_out/<tag>_<what>_kernel_stats.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    c2) ARGS="--config wnet_h256_d36_T200 --steps 20 --warmup 3 --no-extra";;
    c3) ARGS="--config unet_d64_n6_T200 --steps 20 --warmup 3";;
    c4) ARGS="--config unet_d32_n6_T50_cond --steps 20 --warmup 3";;
    d128) ARGS="--config unet_d128_n6_T200 --steps 10 --warmup 2";;
    c5train) ARGS="--config unet_d128_n6_T200 --mode train --steps 4 --warmup 2";;
    wntrain) ARGS="--config wnet_h256_d36_T200 --mode train --steps 6 --warmup 2";;
  esac
  python $R/bench.py $ARGS > /tmp/b_$what.log 2>&1; grep '^{' /tmp/b_$what.log | tail -1 > $OUT/${TAG}_bench_$what.json
  W=/tmp/prof_$what; rm -rf $W; mkdir -p $W
  rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py $ARGS --no-cpu-baseline --no-full-loop > $W/stats.log 2>&1
  python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | head -45 > $OUT/${TAG}_${what}_kernel_stats.txt
  rm -rf $W
done
