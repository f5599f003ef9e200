#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1500 python -m pytest tests/test_sashimi_training_gpu.py -x -q -s -k "bf16x6" 2>&1 | grep -E "d128:|passed|failed|Error|assert" | cut -c1-300 | tail -6
for P in f32 bf16x6; do
 DWS_BENCH_NO_DP_OVERHEAD=1 timeout 900 python bench.py --config unet_d128_n6_T200 --mode train --precision $P --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C5 train $P', round(d['ms_per_step'],2), d['final_loss'])"
done
bash tools/dbg/prof_tr6.sh | grep -E "wgrad|tapconv" | cut -c1-140 | head -6
