#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=/tmp/prof_tr6; rm -rf $W; mkdir -p $W
DWS_BENCH_NO_DP_OVERHEAD=1 rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py --config unet_d128_n6_T200 --mode train --precision bf16x6 --steps 4 --warmup 2 --no-cpu-baseline > $W/stats.log 2>&1
python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | head -40 > $OUT/r05_c5train_bf16x6_kernel_stats.txt
cut -c1-150 $OUT/r05_c5train_bf16x6_kernel_stats.txt | head -32
