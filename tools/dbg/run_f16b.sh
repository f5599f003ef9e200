#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
{ DWS_WINO_TRACE_CHUNKS=1 DWS_BX6_TRACE=1 python tools/wn_layer_times.py --precision f16x3 --reps 1 2>&1 | grep -A10 "trace\] d=1 \|trace\] d=256 " | head -24; } > $O/r05_f16x3_phase_trace.txt
cut -c1-260 $O/r05_f16x3_phase_trace.txt
bash tools/sclk_log.sh $O/r05_sclk_c2_f16x3.txt -- python bench.py --precision f16x3 --steps 200 --warmup 3 --no-cpu-baseline --no-extra --no-roofline --no-full-loop > /dev/null 2>&1
head -3 $O/r05_sclk_c2_f16x3.txt; sed -n 4,12p $O/r05_sclk_c2_f16x3.txt
