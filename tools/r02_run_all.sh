#!/bin/bash
# GPU box: full GPU test suite, then every bench line + rocprof kernel stats + PMC counters of round 2 (-> gpurun_out/r02_*).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/r02_gputest_final.log; tail -3 $OUT/r02_gputest_final.log
tools/sclk_log.sh $OUT/r02_sclk_c2.txt -- python bench.py --config wnet_h256_d36_T200 --steps 40 --warmup 3 --no-cpu-baseline > /tmp/c2clk.log 2>&1
head -3 $OUT/r02_sclk_c2.txt
tools/r02_measure.sh r02 c2 c3 c4 d128 c5train wntrain
tools/profile_kernel.sh r02_sashimi_d64_fftconv fftconv --config unet_d64_n6_T200
tools/profile_kernel.sh r02_sashimi_d64_tail s4_tail --config unet_d64_n6_T200
python - <<'PY'
import json
for w in ('c2','c3','c4','d128','c5train','wntrain'):
    try:
        d=json.load(open('gpurun_out/r02_bench_%s.json'%w)); rf=d.get('roofline',{})
        print(w, round(d['ms_per_step'],3), round(d['value']), rf.get('frac'), rf.get('fftconv',{}).get('frac'), d.get('cpu_baseline',{}).get('value'))
    except Exception as e: print(w, 'ERR', e)
PY
