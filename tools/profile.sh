#!/bin/bash
# Profile the headline bench on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats  -> gpurun_out/prof_stats
#   2. separate --pmc passes (counters only, no trace domains) -> gpurun_out/prof_pmc*
# Summaries are copied into profiles/ by hand after the call.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- $CMD > $OUT/prof_stats.log 2>&1
CMD2="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT -d $OUT/prof_pmc1 -o pmc1 -- $CMD2 > $OUT/prof_pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_pmc2 -o pmc2 -- $CMD2 > $OUT/prof_pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_pmc3 -o pmc3 -- $CMD2 > $OUT/prof_pmc3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES -d $OUT/prof_pmc4 -o pmc4 -- $CMD2 > $OUT/prof_pmc4.log 2>&1
ls -R $OUT | head -50
