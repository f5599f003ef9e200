cd $GRAFT_REPO_ROOT
T="tests/test_sashimi_training_gpu.py::test_sashimi_parameter_gradients_match_autograd"
echo "== default"; python -m pytest "$T" -q -m gpu -s 2>&1 | grep -E "^E  |passed|failed|worst" | cut -c1-200 | head -20
echo "== old ln_bwd"; DWS_LN_BWD_OLD=1 python -m pytest "$T" -q -m gpu -s 2>&1 | grep -E "^E  |passed|failed|worst" | cut -c1-200 | head -20
echo "== tw square"; export DWS_HIPCC_FLAGS_fftconv_kernels="-fno-slp-vectorize -DDWS_FFT_TW_SQUARE"; touch diffwave-sashimi_amd/csrc/fftconv_kernels.hip; python diffwave-sashimi_amd/build.py >/dev/null
python -m pytest "$T" -q -m gpu -s 2>&1 | grep -E "^E  |passed|failed|worst" | cut -c1-200 | head -20
unset DWS_HIPCC_FLAGS_fftconv_kernels; touch diffwave-sashimi_amd/csrc/fftconv_kernels.hip; python diffwave-sashimi_amd/build.py >/dev/null
echo "== ln fusion tests"; python -m pytest tests/test_sashimi_gpu.py -q -m gpu 2>&1 | tail -5 | cut -c1-200
tools/r02_measure.sh r02c c3 c4 d128
python -c "
import json
for w in ('c3','c4','d128'):
    d=json.load(open('gpurun_out/r02c_bench_%s.json'%w)); print(w, d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))
"
head -12 gpurun_out/r02c_c3_kernel_stats.txt | cut -c1-150
