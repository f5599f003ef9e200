cd $GRAFT_REPO_ROOT
for v in 0 3 6 12 18 0; do echo -n "bx3 stagger $v: "; DWS_BX3_STAGGER=$v python bench.py --precision bf16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])'; done
for v in 0 4 8 16 32 0; do echo -n "f32 stagger $v: "; DWS_WN_STAGGER=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])'; done
