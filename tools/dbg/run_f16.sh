#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_f16x3_gpu.py -x -q -s -m gpu > $OUT/f16_tests.log 2>&1; echo "f16 tests rc=$?"
tail -40 $OUT/f16_tests.log
timeout 600 python -m pytest tests/test_bf16x6_gpu.py -x -q -m gpu > $OUT/f16_bx6_tests.log 2>&1; echo "bx6 tests rc=$?"
tail -5 $OUT/f16_bx6_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-full-loop --precision f16x3 > $OUT/f16_bench.json 2> $OUT/f16_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("/root/repo/gpurun_out/f16_bench.json"):
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"], d["value"], d.get("roofline"))
PY
timeout 300 python tools/wn_layer_times.py --precision f16x3 2>&1 | tail -5
timeout 300 python tools/wn_layer_times.py --precision bf16x6 2>&1 | tail -2
