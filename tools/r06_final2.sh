#!/bin/bash
# GPU box, second box of the final build: the new stacked-vs-per-block training test, smoke(), the default bench line and
# rocprofv3 kernel stats of the headline command on the SAME box (the bench's own event timing and rocprof's average must agree).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
SECONDS=0
timeout 900 python -m pytest tests/test_sashimi_training_gpu.py -m gpu -q -s -k "stacked or layernorm_fused" > $O/r06_final2_train_tests.log 2>&1
echo "pytest rc $? after $SECONDS s"; grep -E "stacked vs per-block|passed|failed" $O/r06_final2_train_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_final2_smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/r06_final2_smoke.log
python bench.py > $O/r06_final2_bench_default.json 2> $O/r06_final2_bench_default.err
echo "bench done after $SECONDS s"
( cd /tmp && export TMPDIR=/tmp
  for P in bf16x6 f32; do
    W=/tmp/prof_c2_$P; rm -rf $W; mkdir -p $W
    rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-full-loop --precision $P > $O/r06_final2_bench_c2_${P}_under_rocprof.json 2> $W/stats.log
    python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | head -12 > $O/r06_final2_wavenet_${P}_kernel_stats.txt
    rm -rf $W
  done )
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_final2_bench_*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d.get("roofline") or {}
    print(f, d.get("ms_per_step"), r.get("avg_launch_ms"), r.get("frac"))
    if "summary" in d: print("  summary", json.dumps(d["summary"]))
PY
head -4 $O/r06_final2_wavenet_*_kernel_stats.txt | cut -c1-150
echo "all done after $SECONDS s"
