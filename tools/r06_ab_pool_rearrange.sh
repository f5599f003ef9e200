#!/bin/bash
# GPU box: the vectorized pooling rearrangement (pool_rearrange_vec_kernel) -- training tests that cover it (p = 4 and p = 2,
# forward and adjoint, accumulate / addend), then the same-box A/B of the config-5 training step and rocprofv3 stats of the kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
SECONDS=0
timeout 420 python -m pytest tests/test_sashimi_training_gpu.py tests/test_full_size_gpu.py::test_config5_training_step_at_full_size tests/test_train_cli.py tests/test_learning_gpu.py -m gpu -q -x > $O/r06_pool_vec_tests.log 2>&1
echo "pytest rc $? after $SECONDS s"; tail -3 $O/r06_pool_vec_tests.log
{
echo "# Same-box A/B: pooling rearrangements of the training path, per-element kernel (DWS_POOL_REARRANGE_OLD=1) against"
echo "# pool_rearrange_vec_kernel<P> (a thread owns the P phases of one pooled position: 4P-byte long side, row-contiguous wide side)."
echo "# config-5 training step, B=32; tools/ab_env_plain.sh 2 DWS_POOL_REARRANGE_OLD ...: ms per step, final loss"
bash tools/ab_env_plain.sh 2 DWS_POOL_REARRANGE_OLD --config unet_d128_n6_T200 --mode train --precision f32 --steps 8 --warmup 2
bash tools/ab_env_plain.sh 1 DWS_POOL_REARRANGE_OLD --config unet_d128_n6_T200 --mode train --precision bf16x6 --steps 8 --warmup 2
} > $O/r06_ab_pool_rearrange.txt 2>&1
cat $O/r06_ab_pool_rearrange.txt
( cd /tmp && export TMPDIR=/tmp
  for V in old new; do
    W=/tmp/prof_pool_$V; rm -rf $W; mkdir -p $W
    if [ $V = old ]; then export DWS_POOL_REARRANGE_OLD=1; else unset DWS_POOL_REARRANGE_OLD; fi
    DWS_BENCH_NO_DP_OVERHEAD=1 rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- python $R/bench.py --config unet_d128_n6_T200 --mode train --precision f32 --steps 3 --warmup 1 --no-cpu-baseline > $W/stats.log 2>&1
    echo "## under rocprofv3, $V" >> $O/r06_ab_pool_rearrange.txt
    python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db | grep -i "pool_rearrange" | cut -c1-150 >> $O/r06_ab_pool_rearrange.txt
    rm -rf $W
  done )
tail -5 $O/r06_ab_pool_rearrange.txt
echo "all done after $SECONDS s"
