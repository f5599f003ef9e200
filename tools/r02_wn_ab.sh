#!/bin/bash
# A/B of the headline layer kernel: the next chunk's LDS-DMA issued before (default) or after the first k-group's
# A-fragment request.  gpurun_out/r02_wn_<name>.json = bench lines.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
for v in base: late:-DDWS_WN_DMA_LATE; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_wavenet_kernels="$flags"
  touch diffwave-sashimi_amd/csrc/wavenet_kernels.hip; python diffwave-sashimi_amd/build.py > /dev/null
  python bench.py --config wnet_h256_d36_T200 --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $OUT/r02_wn_$name.json
  python -c "import json;d=json.load(open('$OUT/r02_wn_$name.json'));print('$name', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
unset DWS_HIPCC_FLAGS_wavenet_kernels; touch diffwave-sashimi_amd/csrc/wavenet_kernels.hip; python diffwave-sashimi_amd/build.py > /dev/null
