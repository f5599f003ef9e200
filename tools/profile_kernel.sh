#!/bin/bash
# usage: tools/profile_kernel.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>_{stats,pmcN}
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline $*"
rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_stats -o stats -- $CMD > $OUT/prof_${TAG}_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d $OUT/prof_${TAG}_pmc1 -o pmc1 -- $CMD > $OUT/prof_${TAG}_pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_${TAG}_pmc2 -o pmc2 -- $CMD > $OUT/prof_${TAG}_pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_${TAG}_pmc3 -o pmc3 -- $CMD > $OUT/prof_${TAG}_pmc3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_MFMA -d $OUT/prof_${TAG}_pmc4 -o pmc4 -- $CMD > $OUT/prof_${TAG}_pmc4.log 2>&1
