#!/bin/bash
# start-skew experiments (DWS_WN_STAGGER / DWS_WN_STAGGER2 / DWS_BX3_STAGGER), same box
cd $GRAFT_REPO_ROOT
for v in 0 64 128 256 400 527 0; do echo -n "f32 second-WG skew $v: "; DWS_WN_STAGGER2=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])'; done
