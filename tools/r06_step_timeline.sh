#!/bin/bash
# Timeline of one C5 training step (launch sequence, idle gaps, small launches):  tools/r06_step_timeline.sh [precision]
R=${GRAFT_REPO_ROOT:-/root/repo}
PREC=${1:-f32}
cd /tmp && export TMPDIR=/tmp
export DWS_BENCH_NO_DP_OVERHEAD=1
W=/tmp/tl; rm -rf $W; mkdir -p $W $R/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $W -o t -- python $R/bench.py --config unet_d128_n6_T200 --mode train --precision $PREC \
   --steps 3 --warmup 2 --no-cpu-baseline --no-extra --no-full-loop --no-roofline > $W/log 2>&1
grep '^{' $W/log | tail -1 | cut -c1-400
CSV=$(find $W -name '*kernel_trace.csv' | head -1)
python $R/tools/step_timeline.py $CSV 3 --seq > $R/gpurun_out/step_timeline_$PREC.txt
head -80 $R/gpurun_out/step_timeline_$PREC.txt
