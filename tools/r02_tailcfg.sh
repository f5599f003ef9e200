#!/bin/bash
# tail kernel tile-shape sweep, same box: build variants x DWS_TAIL_CFG x config
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for fl in "" "-DDWS_TAIL_KG_UNROLL"; do
export DWS_HIPCC_FLAGS_sashimi_mfma="$fl"; touch $R/diffwave-sashimi_amd/csrc/sashimi_mfma.hip; python $R/diffwave-sashimi_amd/build.py > /dev/null
for cfg in unet_d64_n6_T200 unet_d32_n6_T50_cond; do for alt in 0 1; do
  W=/tmp/prof_t$alt; rm -rf $W; mkdir -p $W
  DWS_TAIL_CFG=$alt rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $W/log 2>&1
  echo "== [$fl] $cfg alt $alt: $(grep '^{' $W/log | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | grep s4_tail | cut -c1-150
  rm -rf $W
done; done; done
unset DWS_HIPCC_FLAGS_sashimi_mfma; touch $R/diffwave-sashimi_amd/csrc/sashimi_mfma.hip; python $R/diffwave-sashimi_amd/build.py > /dev/null
