#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "vec16:" "dword:-DDWS_WINO_DWORD_EPI" "vec16:" "dword:-DDWS_WINO_DWORD_EPI"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_wavenet_wino="$flags"
  python diffwave-sashimi_amd/build.py > /dev/null 2>&1
  echo "== $name: $(python tools/wn_layer_times.py --reps 5 2>&1 | tail -1)"
  python bench.py --steps 40 --no-cpu-baseline --no-extra --no-roofline --no-full-loop 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('   bench ms/step', d['ms_per_step'])"
done
export DWS_HIPCC_FLAGS_wavenet_wino=""
python diffwave-sashimi_amd/build.py > /dev/null 2>&1
python -m pytest tests/test_wavenet_gpu.py tests/test_sampler_gpu.py tests/test_wavenet_training_gpu.py -q -x 2>&1 | tail -3
DWS_WINO_TRACE=1 python tools/wn_layer_times.py --reps 1 2>&1 | grep "d=256 " | head -1
