cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sashimi_gpu.py tests/test_full_size_gpu.py -x -q --timeout 600 2>&1 | tail -2
timeout 900 tools/ab_lib.sh "fftconv_kernel<1" --config unet_d64_n6_T200 --steps 60 --warmup 5 --no-roofline 2>&1 | tail -16
