#!/bin/bash
# GPU box: round-4 closing run after the fused S4-convolution tail: the full GPU suite, the default bench line, the SaShiMi
# bench lines + kernel stats, the convolution kernel's counters and phase trace (-> gpurun_out/r04_*).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=8 2>&1 | tail -30 > $OUT/r04_gputest.log; tail -3 $OUT/r04_gputest.log
T0=$(date +%s); timeout 600 python bench.py > /tmp/b_default.log 2> /tmp/b_default.err; T1=$(date +%s)
grep '^{' /tmp/b_default.log | tail -1 > $OUT/r04_bench_default.json
echo "python bench.py: $((T1 - T0)) s wall clock" > $OUT/r04_bench_default_wallclock.txt
timeout 900 tools/r02_measure.sh r04 c3 c4 d128 c5train
timeout 600 tools/profile_kernel.sh r04_sashimi_d64_fftconv fftconv --config unet_d64_n6_T200
timeout 300 python tools/fft_trace.py unet_d64_n6_T200 2> $OUT/r04_fft_trace_c3.txt > /dev/null
python - <<'PY'
import json
for w in ('default','c3','c4','d128','c5train'):
    try:
        d=json.load(open('gpurun_out/r04_bench_%s.json'%w)); rf=d.get('roofline',{})
        print(w, round(d['ms_per_step'],3), round(d['value']), rf.get('frac'), rf.get('fftconv',{}).get('frac'), d.get('cpu_baseline',{}).get('value'), (d.get('full_loop') or {}).get('ratio_to_timed_ms_per_step'), (d.get('dp') or {}).get('dp_overhead_ms'))
        if 'extra_configs' in d:
            for k,v in d['extra_configs'].items(): print('   extra', k, round(v.get('ms_per_step',0),3), (v.get('roofline') or {}).get('frac'), v.get('error'))
    except Exception as e: print(w, 'ERR', e)
PY
