#!/bin/bash
# GPU box: the part of tools/r04_run_all.sh after the per-config bench lines (default line, counters, traces).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
T0=$(date +%s); python bench.py > /tmp/b_default.log 2> /tmp/b_default.err; T1=$(date +%s)
grep '^{' /tmp/b_default.log | tail -1 > $OUT/r04_bench_default.json
echo "python bench.py: $((T1 - T0)) s wall clock" > $OUT/r04_bench_default_wallclock.txt
tail -3 /tmp/b_default.err
tools/r04_traffic.sh r04 > /dev/null
tools/profile_kernel.sh r04_sashimi_d64_fftconv fftconv --config unet_d64_n6_T200
tools/profile_kernel.sh r04_sashimi_d64_tail s4_tail --config unet_d64_n6_T200
python tools/wn_layer_times.py --reps 5 > $OUT/r04_wino_layer_times.txt 2>/dev/null
python tools/tail_trace.py unet_d64_n6_T200 2> $OUT/r04_tail_phase_trace.txt > /dev/null
DWS_WINO_TRACE_CHUNKS=1 DWS_WINO_TRACE=1 python tools/wn_layer_times.py --reps 1 2>&1 | grep -A10 "d=256 " | head -11 > $OUT/r04_wino_phase_trace.txt
ls -la $OUT/r04_* | awk '{print $5, $9}'
