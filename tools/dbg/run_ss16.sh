#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_sashimi_bf16x6_gpu.py -x -q -s 2>&1 | grep -E "vs float64|passed|failed|Error|error" | cut -c1-330
for C in unet_d64_n6_T200 unet_d32_n6_T50_cond; do for P in f32 bf16x6 f16x3; do
 timeout 600 python bench.py --config $C --precision $P --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-full-loop --no-roofline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$C $P', round(d['ms_per_step'],3), d['dtype'][:40])"
done; done
