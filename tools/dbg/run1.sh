cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DWS_BENCH_NO_DP_OVERHEAD=1
for pf in 13 12 10 13 12 10; do
  export DWS_FFT_PERSIST_FROM=$pf
  W=/tmp/ab_p; rm -rf $W; mkdir -p $W
  rocprofv3 --kernel-trace --stats -d $W -o s -- python $R/bench.py --config unet_d64_n6_T200 --steps 60 --warmup 5 --no-cpu-baseline --no-extra --no-full-loop --no-roofline > $W/log 2>&1
  echo "== persist_from=$pf: $(grep '^{' $W/log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  python $R/tools/rocpd_summary.py stats $W/s_results.db | grep -i "fftconv_kernel<1" | cut -c1-160
done
