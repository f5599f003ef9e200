#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_sashimi_bf16x6_gpu.py -x -q -s 2>&1 | grep -E "vs float64|passed|failed|Error|error" | cut -c1-330
for C in unet_d64_n6_T200 unet_d128_n6_T200; do for P in f32 bf16x6 f16x3; do
 timeout 600 python bench.py --config $C --precision $P --steps 20 --warmup 4 --no-cpu-baseline --no-extra --no-full-loop 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$C $P', round(d['ms_per_step'],3), 'tails frac', round(d['roofline']['frac'],3), 'tails ms', round(d['roofline']['ms_per_step_in_kernel'],3))"
done; done
DWS_TAIL_NO_SPLIT_TILE=1 timeout 600 python bench.py --config unet_d64_n6_T200 --precision f16x3 --steps 20 --warmup 4 --no-cpu-baseline --no-extra --no-full-loop --no-roofline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C3 f16x3 without the split tile kernel', round(d['ms_per_step'],3))"
