#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_bf16x6_gpu.py -q 2>&1 | grep -E "passed|failed" | cut -c1-250 | head -5
for v in "nt:" "nont:-DBX6_ABL_NO_NT"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_wavenet_bx6="$flags"
  python diffwave-sashimi_amd/build.py > /dev/null 2>&1
  echo "== $name"
  python tools/wn_layer_times.py --precision bf16x6 --reps 3 2>&1 | tail -1
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf; 
  rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o p -- python $R/bench.py --precision bf16x6 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-full-loop > /dev/null 2>&1
  python $R/tools/rocpd_summary.py pmc /tmp/pf/p_results.db wn_layer_bx6 | cut -c60-200; cd $R
done
