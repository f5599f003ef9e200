#!/bin/bash
# A/B of fftconv build variants on the GPU box: tools/r02_fft_ab.sh name1:"flags" name2:"flags" ...
# per-kernel averages from rocprofv3 on the C3 bench -> gpurun_out/r02_fft_<name>_stats.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_fftconv_kernels="$flags"
  touch $R/diffwave-sashimi_amd/csrc/fftconv_kernels.hip
  python $R/diffwave-sashimi_amd/build.py > /dev/null
  W=/tmp/prof_$name; rm -rf $W; mkdir -p $W
  rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py --config unet_d64_n6_T200 --steps 10 --warmup 2 --no-cpu-baseline > $W/stats.log 2>&1
  python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | grep -E "fftconv|kernel " > $OUT/r02_fft_${name}_stats.txt
  grep '^{' $W/stats.log | tail -1 > $OUT/r02_fft_${name}_bench.json
  rm -rf $W
done
unset DWS_HIPCC_FLAGS_fftconv_kernels
touch $R/diffwave-sashimi_amd/csrc/fftconv_kernels.hip; python $R/diffwave-sashimi_amd/build.py > /dev/null
