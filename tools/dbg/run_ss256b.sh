#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for C in unet_d64_n6_T200 unet_d128_n6_T200; do for P in bf16x6 f16x3; do for E in 0 1; do
 if [ $E = 1 ]; then export DWS_TAIL_NO_SPLIT_TILE=1; else unset DWS_TAIL_NO_SPLIT_TILE; fi
 timeout 600 python bench.py --config $C --precision $P --steps 30 --warmup 4 --no-cpu-baseline --no-extra --no-full-loop --no-roofline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$C $P no_split_tile=$E', round(d['ms_per_step'],3))"
done; done; done
