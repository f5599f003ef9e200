#!/bin/bash
# GPU box, round 5: every number DESIGN.md section 6 and profiles/README.md quote.  tools/r05_run_all.sh [part]
#   part 1: bench lines (default run, per-config legs at f32 and bf16x6, training)      -> gpurun_out/r05_bench_*.json
#   part 2: rocprofv3 kernel stats + PMC passes, counter traffic, phase traces, clocks  -> gpurun_out/r05_*.txt / .json
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
PART=${1:-1}
if [ "$PART" = 1 ]; then
  SECONDS=0
  python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err
  echo "$SECONDS s wall (python bench.py, all legs)" > $O/r05_bench_default_wallclock.txt
  python bench.py --precision bf16x6 --no-extra > $O/r05_bench_c2_bf16x6.json 2>> $O/r05_bench_default.err
  python bench.py --precision f16x3 --no-extra --no-cpu-baseline > $O/r05_bench_c2_f16x3.json 2>> $O/r05_bench_default.err
  for C in unet_d64_n6_T200 unet_d32_n6_T50_cond; do
    python bench.py --config $C --no-extra --no-cpu-baseline > $O/r05_bench_${C}.json 2>> $O/r05_bench_default.err
    python bench.py --config $C --no-extra --no-cpu-baseline --precision bf16x6 > $O/r05_bench_${C}_bf16x6.json 2>> $O/r05_bench_default.err
    python bench.py --config $C --no-extra --no-cpu-baseline --precision f16x3 > $O/r05_bench_${C}_f16x3.json 2>> $O/r05_bench_default.err
  done
  python bench.py --config unet_d128_n6_T200 --mode train --steps 6 --warmup 2 > $O/r05_bench_c5train.json 2>> $O/r05_bench_default.err
  DWS_BENCH_NO_DP_OVERHEAD=1 python bench.py --config unet_d128_n6_T200 --mode train --precision bf16x6 --steps 6 --warmup 2 > $O/r05_bench_c5train_bf16x6.json 2>> $O/r05_bench_default.err
  python bench.py --config wnet_h128_d30_T200 --no-extra --no-cpu-baseline > $O/r05_bench_wavenet_h128.json 2>> $O/r05_bench_default.err
  python bench.py --config wnet_h128_d30_T200 --no-extra --no-cpu-baseline --precision bf16x6 > $O/r05_bench_wavenet_h128_bf16x6.json 2>> $O/r05_bench_default.err
  python bench.py --config wnet_h128_d30_T200 --no-extra --no-cpu-baseline --precision f16x3 > $O/r05_bench_wavenet_h128_f16x3.json 2>> $O/r05_bench_default.err
  for f in $O/r05_bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][0])
except Exception as e:
    print("unreadable:", e); sys.exit(0)
print({k: d.get(k) for k in ("value", "ms_per_step", "dtype")}, "roofline frac", (d.get("roofline") or {}).get("frac"))
for k in ("extra_bf16x6", "extra_f16x3", "extra_bf16x3"):
    if k in d: print(" ", k, d[k].get("ms_per_step"), (d[k].get("roofline") or {}).get("frac"))
for k, v in (d.get("extra_configs") or {}).items():
    print(" ", k, v.get("ms_per_step"), v.get("error"), (v.get("extra_bf16x6") or {}).get("ms_per_step"), (v.get("extra_f16x3") or {}).get("ms_per_step"))
if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("cores"), (d["cpu_baseline"].get("whole_host") or {}))
PY
  done
else
  bash tools/profile_kernel.sh r05_wavenet_f32 wn_layer
  bash tools/profile_kernel.sh r05_wavenet_bf16x6 wn_layer --precision bf16x6
  bash tools/profile_kernel.sh r05_wavenet_f16x3 wn_layer --precision f16x3
  bash tools/profile_kernel.sh r05_sashimi_d64_tail s4_tail --config unet_d64_n6_T200
  bash tools/profile_kernel.sh r05_sashimi_d64_tail_bf16x6 s4_tail --config unet_d64_n6_T200 --precision bf16x6
  bash tools/profile_kernel.sh r05_sashimi_d32_tail_bf16x6 s4_tail --config unet_d32_n6_T50_cond --precision bf16x6
  bash tools/profile_kernel.sh r05_sashimi_d64_tail_f16x3 s4_tail --config unet_d64_n6_T200 --precision f16x3
  bash tools/r05_traffic.sh r05 f32 > /dev/null
  bash tools/r05_traffic.sh r05 bf16x6 > /dev/null
  bash tools/r05_traffic.sh r05 f16x3 > /dev/null
  bash tools/r05_traffic_sashimi.sh unet_d64_n6_T200 r05 > /dev/null
  bash tools/r05_traffic_sashimi.sh unet_d32_n6_T50_cond r05 > /dev/null
  cd $R
  { DWS_WINO_TRACE_CHUNKS=1 DWS_BX6_TRACE=1 python tools/wn_layer_times.py --precision bf16x6 --reps 1 2>&1 | grep -A10 "trace\] d=1 \|trace\] d=256 " | head -24; } > $O/r05_bx6_phase_trace.txt
  { DWS_WINO_TRACE_CHUNKS=1 DWS_BX6_TRACE=1 python tools/wn_layer_times.py --precision f16x3 --reps 1 2>&1 | grep -A10 "trace\] d=1 \|trace\] d=256 " | head -24; } > $O/r05_f16x3_phase_trace.txt
  { python tools/wn_layer_times.py --precision f16x3 --reps 5; python tools/wn_layer_times.py --precision bf16x6 --reps 5; python tools/wn_layer_times.py --precision f32 --reps 5; } > $O/r05_wn_layer_times.txt 2>&1
  { for C in unet_d64_n6_T200 unet_d32_n6_T50_cond; do for P in f32 bf16x6 f16x3; do echo "== $C $P"; python tools/tail_trace.py $C $P 2>&1 | grep "chain" | sort | uniq -c | sort -rn | tail -4; done; done; } > $O/r05_chain_phase_trace.txt
  bash tools/sclk_log.sh $O/r05_sclk_c2_bf16x6.txt -- python bench.py --precision bf16x6 --steps 150 --warmup 3 --no-cpu-baseline --no-extra --no-roofline --no-full-loop > /dev/null 2>&1
  bash tools/sclk_log.sh $O/r05_sclk_c2_f16x3.txt -- python bench.py --precision f16x3 --steps 200 --warmup 3 --no-cpu-baseline --no-extra --no-roofline --no-full-loop > /dev/null 2>&1
  bash tools/sclk_log.sh $O/r05_sclk_c2_f32.txt -- python bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-extra --no-roofline --no-full-loop > /dev/null 2>&1
  bash tools/r02_measure.sh r05 c5train > /dev/null 2>&1
  bash tools/dbg/prof_tr6.sh > /dev/null 2>&1
  ls -la $O | grep r05_ | awk '{print $5, $9}'
fi
