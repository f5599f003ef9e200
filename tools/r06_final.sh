#!/bin/bash
# GPU box, end of round 6: the numbers of the FINAL build in one call (bench lines first, so they survive a clamped call).
#   tools/r06_final.sh            bench default line, C5 training lines + kernel stats (both precisions), full GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
SECONDS=0
python bench.py > $O/r06_final_bench_default.json 2> $O/r06_final_bench_default.err
echo "$SECONDS s wall (python bench.py, all legs)" > $O/r06_final_bench_default_wallclock.txt
python bench.py --config unet_d128_n6_T200 --mode train --precision f32 --steps 6 --warmup 2 > $O/r06_final_bench_c5train_f32.json 2>> $O/r06_final_bench_default.err
DWS_BENCH_NO_DP_OVERHEAD=1 python bench.py --config unet_d128_n6_T200 --mode train --precision bf16x6 --steps 6 --warmup 2 > $O/r06_final_bench_c5train_bf16x6.json 2>> $O/r06_final_bench_default.err
( cd /tmp && export TMPDIR=/tmp
  for P in f32 bf16x6; do
    W=/tmp/prof_c5_$P; rm -rf $W; mkdir -p $W
    DWS_BENCH_NO_DP_OVERHEAD=1 rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py --config unet_d128_n6_T200 --mode train --precision $P --steps 4 --warmup 2 --no-cpu-baseline > $W/stats.log 2>&1
    python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | head -60 > $O/r06_final_c5train_${P}_kernel_stats.txt
    rm -rf $W
  done )
for f in $O/r06_final_bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][0])
except Exception as e:
    print("unreadable:", e); sys.exit(0)
print({k: d.get(k) for k in ("value", "ms_per_step", "dtype")}, "roofline frac", (d.get("roofline") or {}).get("frac"))
if "summary" in d: print("  summary", json.dumps(d["summary"]))
PY
done
echo "bench part: $SECONDS s"
timeout 1500 python -m pytest tests -m gpu -q > $O/r06_final_gputest.log 2>&1
echo "pytest rc $? after $SECONDS s"; tail -3 $O/r06_final_gputest.log
