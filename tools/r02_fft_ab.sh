#!/bin/bash
# A/B of fftconv codegen on the GPU box: SLP-vectorised (v_pk_*) vs scalar, per-kernel averages from rocprofv3.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for variant in scalar slp; do
  if [ $variant = slp ]; then export DWS_HIPCC_FLAGS_fftconv_kernels=" "; else unset DWS_HIPCC_FLAGS_fftconv_kernels; fi
  touch $R/diffwave-sashimi_amd/csrc/fftconv_kernels.hip
  python $R/diffwave-sashimi_amd/build.py > /dev/null
  W=/tmp/prof_$variant; rm -rf $W; mkdir -p $W
  rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py --config unet_d64_n6_T200 --steps 10 --warmup 2 --no-cpu-baseline > $W/stats.log 2>&1
  python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | head -14 > $OUT/r02_fft_${variant}_stats.txt
  grep '^{' $W/stats.log | tail -1 > $OUT/r02_fft_${variant}_bench.json
done
unset DWS_HIPCC_FLAGS_fftconv_kernels
touch $R/diffwave-sashimi_amd/csrc/fftconv_kernels.hip; python $R/diffwave-sashimi_amd/build.py > /dev/null
