#!/bin/bash
# GPU box: every bench line + rocprof kernel stats + PMC counters of round 3 (-> gpurun_out/r03_*).
#   tools/r03_run_all.sh [tests]     ("tests": the full GPU suite first)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
if [ "$1" = "tests" ]; then
  python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 > $OUT/r03_gputest.log; tail -3 $OUT/r03_gputest.log
fi
# the line the driver records: default command (headline + extra_configs + cpu_baseline)
python bench.py > /tmp/b_default.log 2>&1; grep '^{' /tmp/b_default.log | tail -1 > $OUT/r03_bench_default.json
tools/sclk_log.sh $OUT/r03_sclk_c2.txt -- python bench.py --config wnet_h256_d36_T200 --steps 40 --warmup 3 --no-cpu-baseline --no-extra > /tmp/c2clk.log 2>&1
head -3 $OUT/r03_sclk_c2.txt
tools/r02_measure.sh r03 c2 c3 c4 d128 c5train wntrain
tools/profile_kernel.sh r03_wavenet_f32 wn_layer_wino
DWS_WN_DIRECT=1 python bench.py --no-cpu-baseline --no-extra --steps 20 2>/dev/null | grep '^{' | tail -1 > $OUT/r03_bench_c2_direct.json
tools/profile_kernel.sh r03_sashimi_d64_fftconv fftconv --config unet_d64_n6_T200
tools/profile_kernel.sh r03_sashimi_d64_tail s4_tail --config unet_d64_n6_T200
python tools/wn_layer_times.py --reps 5 > $OUT/r03_wino_layer_times.txt 2>/dev/null
DWS_WINO_TRACE_CHUNKS=1 DWS_WINO_TRACE=1 python tools/wn_layer_times.py --reps 1 2>&1 | grep -A10 "d=256 " | head -11 > $OUT/r03_wino_phase_trace.txt
python - <<'PY'
import json
for w in ('default','c2','c2_direct','c3','c4','d128','c5train','wntrain'):
    try:
        d=json.load(open('gpurun_out/r03_bench_%s.json'%w)); rf=d.get('roofline',{})
        print(w, round(d['ms_per_step'],3), round(d['value']), rf.get('frac'), rf.get('fftconv',{}).get('frac'), d.get('cpu_baseline',{}).get('value'))
        if 'extra_configs' in d:
            for k,v in d['extra_configs'].items(): print('   extra', k, round(v['ms_per_step'],3), v['roofline'].get('frac'))
    except Exception as e: print(w, 'ERR', e)
PY
