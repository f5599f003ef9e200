#!/bin/bash
# first GPU contact of the bf16x6 layer: accuracy tests, per-dilation times, phase trace, short bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_bf16x6_gpu.py -x -q -s 2>&1 | tail -40 > $O/r05_bx6_tests.log
python tools/wn_layer_times.py --precision bf16x6 --reps 3 > $O/r05_bx6_layer_times.txt 2>&1
python tools/wn_layer_times.py --precision f32 --reps 3 >> $O/r05_bx6_layer_times.txt 2>&1
DWS_WINO_TRACE_CHUNKS=1 DWS_BX6_TRACE=1 python tools/wn_layer_times.py --precision bf16x6 --reps 1 2>&1 | grep -A10 "d=1 \|d=256 " | head -40 > $O/r05_bx6_trace.txt
python bench.py --precision bf16x6 --no-cpu-baseline --no-extra --steps 20 > $O/r05_bx6_bench.json 2> $O/r05_bx6_bench.err
tail -5 $O/r05_bx6_tests.log; cat $O/r05_bx6_layer_times.txt | tail -30; cat $O/r05_bx6_trace.txt | head -24; cat $O/r05_bx6_bench.json | cut -c1-1500
