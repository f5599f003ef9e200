#!/bin/bash
# GPU box, round 6: every number DESIGN.md section 6 and profiles/README.md quote.  tools/r06_run_all.sh [part]
#   part 1: bench lines (default run = bf16x6 headline with the exact-f32 legs beside it; per-config lines; training)
#   part 2: rocprofv3 kernel stats + PMC passes of the dominant kernels, counter traffic, clocks / board power
#   part 3: FETCH_SIZE calibration, phase traces
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
PART=${1:-1}
if [ "$PART" = 1 ]; then
  SECONDS=0
  python bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err
  echo "$SECONDS s wall (python bench.py, all legs)" > $O/r06_bench_default_wallclock.txt
  python bench.py --precision f32 --no-extra --no-cpu-baseline > $O/r06_bench_c2_f32.json 2>> $O/r06_bench_default.err
  for C in unet_d64_n6_T200 unet_d32_n6_T50_cond wnet_h128_d30_T200; do
    for P in bf16x6 f32; do
      python bench.py --config $C --no-extra --no-cpu-baseline --precision $P > $O/r06_bench_${C}_$P.json 2>> $O/r06_bench_default.err
    done
  done
  python bench.py --config unet_d128_n6_T200 --mode train --precision f32 --steps 6 --warmup 2 > $O/r06_bench_c5train_f32.json 2>> $O/r06_bench_default.err
  DWS_BENCH_NO_DP_OVERHEAD=1 python bench.py --config unet_d128_n6_T200 --mode train --precision bf16x6 --steps 6 --warmup 2 > $O/r06_bench_c5train_bf16x6.json 2>> $O/r06_bench_default.err
  for f in $O/r06_bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][0])
except Exception as e:
    print("unreadable:", e); sys.exit(0)
print({k: d.get(k) for k in ("value", "ms_per_step", "dtype")}, "roofline frac", (d.get("roofline") or {}).get("frac"))
if "summary" in d: print("  summary", json.dumps(d["summary"]))
PY
  done
elif [ "$PART" = 2 ]; then
  bash tools/profile_kernel.sh r06_wavenet_bf16x6 wn_layer --precision bf16x6
  bash tools/profile_kernel.sh r06_wavenet_f32 wn_layer --precision f32
  bash tools/profile_kernel.sh r06_sashimi_d64_tail_bf16x6 s4_tail --config unet_d64_n6_T200 --precision bf16x6
  bash tools/profile_kernel.sh r06_sashimi_d64_tail_f32 s4_tail --config unet_d64_n6_T200 --precision f32
  bash tools/profile_kernel.sh r06_sashimi_d32_tail_bf16x6 s4_tail --config unet_d32_n6_T50_cond --precision bf16x6
  bash tools/r05_traffic.sh r06 bf16x6 > /dev/null
  bash tools/r05_traffic.sh r06 f32 > /dev/null
  for P in bf16x6 f32; do
    bash tools/r06_traffic_sashimi.sh unet_d64_n6_T200 r06 $P > /dev/null
    bash tools/r06_traffic_sashimi.sh unet_d32_n6_T50_cond r06 $P > /dev/null
  done
  cd $R
  bash tools/sclk_log.sh $O/r06_sclk_c2_bf16x6.txt -- python bench.py --precision bf16x6 --steps 150 --warmup 3 --no-cpu-baseline --no-extra --no-roofline --no-full-loop > /dev/null 2>&1
  bash tools/sclk_log.sh $O/r06_sclk_c2_f32.txt -- python bench.py --precision f32 --steps 100 --warmup 3 --no-cpu-baseline --no-extra --no-roofline --no-full-loop > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  for P in f32 bf16x6; do
    W=/tmp/prof_c5_$P; rm -rf $W; mkdir -p $W
    DWS_BENCH_NO_DP_OVERHEAD=1 rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py --config unet_d128_n6_T200 --mode train --precision $P --steps 4 --warmup 2 --no-cpu-baseline > $W/stats.log 2>&1
    python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | head -45 > $O/r06_c5train_${P}_kernel_stats.txt
    rm -rf $W
  done
  ls -la $O | grep r06_ | awk '{print $5, $9}'
else
  bash tools/r06_fetch_calib.sh
  cd $R
  { DWS_WINO_TRACE_CHUNKS=1 DWS_BX6_TRACE=1 python tools/wn_layer_times.py --precision bf16x6 --reps 1 2>&1 | grep -A10 "trace\] d=1 \|trace\] d=256 " | head -24; } > $O/r06_bx6_phase_trace.txt
  python tools/fft_trace.py unet_d64_n6_T200 > $O/r06_fft_trace_c3.txt 2>&1
fi
