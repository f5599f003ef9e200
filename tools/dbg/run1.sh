cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_sashimi_gpu.py tests/test_full_size_gpu.py tests/test_sashimi_training_gpu.py tests/test_sampler_gpu.py -x -q --timeout 600 2>&1 | tail -3
timeout 300 python tools/fft_trace.py unet_d64_n6_T200 2> gpurun_out/r04_fft_trace_c3_fused.txt; grep "M=16384" gpurun_out/r04_fft_trace_c3_fused.txt | tail -4 | cut -c1-420; grep "M=4096" gpurun_out/r04_fft_trace_c3_fused.txt | tail -1 | cut -c1-420
echo "#### C3 DWS_FFT_NO_FUSED_TAIL"
timeout 600 tools/ab_env.sh DWS_FFT_NO_FUSED_TAIL fftconv_kernel --config unet_d64_n6_T200 --steps 60 --warmup 5 2>&1 | tail -14
