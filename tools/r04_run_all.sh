#!/bin/bash
# GPU box: every bench line + rocprof kernel stats + PMC counters of round 4 (-> gpurun_out/r04_*).
#   tools/r04_run_all.sh [tests]     ("tests": the full GPU suite first)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
if [ "$1" = "tests" ]; then
  python -m pytest tests -m gpu -q --timeout 900 --durations=8 2>&1 | tail -30 > $OUT/r04_gputest.log; tail -3 $OUT/r04_gputest.log
fi
# the line the driver records: default command (headline + full loop + extra_configs + cpu_baseline)
T0=$(date +%s); python bench.py > /tmp/b_default.log 2> /tmp/b_default.err; T1=$(date +%s)
grep '^{' /tmp/b_default.log | tail -1 > $OUT/r04_bench_default.json
echo "python bench.py: $((T1 - T0)) s wall clock" > $OUT/r04_bench_default_wallclock.txt
# clock under the Winograd kernel ALONE (no bf16x3 leg, no roofline leg: 400 replays of the f32 step)
tools/sclk_log.sh $OUT/r04_sclk_c2_f32_only.txt -- python bench.py --config wnet_h256_d36_T200 --steps 400 --warmup 3 --no-cpu-baseline --no-extra --no-roofline > /tmp/c2clk.log 2>&1
head -3 $OUT/r04_sclk_c2_f32_only.txt
tools/r02_measure.sh r04 c2 c3 c4 d128 c5train wntrain
tools/profile_kernel.sh r04_wavenet_f32 wn_layer_wino
tools/r04_traffic.sh r04 > /dev/null
tools/profile_kernel.sh r04_sashimi_d64_fftconv fftconv --config unet_d64_n6_T200
tools/profile_kernel.sh r04_sashimi_d64_tail s4_tail --config unet_d64_n6_T200
python tools/wn_layer_times.py --reps 5 > $OUT/r04_wino_layer_times.txt 2>/dev/null
python tools/tail_trace.py unet_d64_n6_T200 2> $OUT/r04_tail_phase_trace.txt > /dev/null
python - <<'PY'
import json
for w in ('default','c2','c3','c4','d128','c5train','wntrain'):
    try:
        d=json.load(open('gpurun_out/r04_bench_%s.json'%w)); rf=d.get('roofline',{})
        print(w, round(d['ms_per_step'],3), round(d['value']), rf.get('frac'), rf.get('fftconv',{}).get('frac'), d.get('cpu_baseline',{}).get('value'), (d.get('full_loop') or {}).get('ratio_to_timed_ms_per_step'), (d.get('dp') or {}).get('dp_overhead_ms'))
        if 'extra_configs' in d:
            for k,v in d['extra_configs'].items(): print('   extra', k, round(v.get('ms_per_step',0),3), (v.get('roofline') or {}).get('frac'), v.get('error'))
    except Exception as e: print(w, 'ERR', e)
PY
