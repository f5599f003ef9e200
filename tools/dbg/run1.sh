cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_sashimi_gpu.py -x -q --timeout 600 2>&1 | tail -2
timeout 300 python tools/fft_trace.py unet_d64_n6_T200 2> gpurun_out/r04_fft_trace_c3_early.txt; grep "M=16384" gpurun_out/r04_fft_trace_c3_early.txt | tail -4 | cut -c1-420
timeout 900 tools/ab_lib.sh "fftconv_kernel<14" --config unet_d64_n6_T200 --steps 60 --warmup 5 --no-roofline 2>&1 | tail -10
