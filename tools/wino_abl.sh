#!/bin/bash
# same-box ablations of the Winograd layer kernel: tools/wino_abl.sh name1:"-DFLAG" name2:"..." ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  export DWS_HIPCC_FLAGS_wavenet_wino="$flags"
  touch $R/diffwave-sashimi_amd/csrc/wavenet_wino.hip; python $R/diffwave-sashimi_amd/build.py > /dev/null
  echo "== $name ($flags): $(python $R/tools/wn_layer_times.py --reps 3 $WINO_ABL_ARGS 2>/dev/null | tail -1)"
  DWS_WINO_TRACE_CHUNKS=1 DWS_WINO_TRACE=1 python $R/tools/wn_layer_times.py --reps 1 2>&1 | grep -A2 "d=256 " | head -3
done
unset DWS_HIPCC_FLAGS_wavenet_wino
touch $R/diffwave-sashimi_amd/csrc/wavenet_wino.hip; python $R/diffwave-sashimi_amd/build.py > /dev/null
