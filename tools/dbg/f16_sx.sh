#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for sx in 16.f 4.f; do
export DWS_HIPCC_FLAGS_wavenet_bx6="-DF16X3_SX=$sx"
python diffwave-sashimi_amd/build.py > /dev/null 2>&1
echo "== SX=$sx"
python -m pytest tests/test_f16x3_gpu.py -q -s -k "float64" 2>&1 | grep -E "f16x3|passed|failed" | cut -c1-330
done
