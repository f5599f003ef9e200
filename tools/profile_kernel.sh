#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_kernel.sh <tag> <kernel-substring> <bench args...>
# Runs bench.py under rocprofv3 (kernel-trace stats, then SEPARATE --pmc passes as the guide prescribes),
# summarises the rocpd databases to text under gpurun_out/ and deletes the databases (they are large).
set -u
TAG=$1; KSUB=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=/tmp/prof_$TAG; rm -rf $W; mkdir -p $W
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-full-loop $*"
CMD2="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-full-loop $*"
rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- $CMD > $W/stats.log 2>&1
python $R/tools/rocpd_summary.py stats $W/stats/stats_results.db > $OUT/${TAG}_kernel_stats.txt
{
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d $W/pmc1 -o pmc1 -- $CMD2 > $W/pmc1.log 2>&1
python $R/tools/rocpd_summary.py pmc $W/pmc1/pmc1_results.db "$KSUB"; rm -rf $W/pmc1
rocprofv3 --pmc FETCH_SIZE -d $W/pmc2 -o pmc2 -- $CMD2 > $W/pmc2.log 2>&1
python $R/tools/rocpd_summary.py pmc $W/pmc2/pmc2_results.db "$KSUB"; rm -rf $W/pmc2
rocprofv3 --pmc WRITE_SIZE -d $W/pmc3 -o pmc3 -- $CMD2 > $W/pmc3.log 2>&1
python $R/tools/rocpd_summary.py pmc $W/pmc3/pmc3_results.db "$KSUB"; rm -rf $W/pmc3
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_MFMA -d $W/pmc4 -o pmc4 -- $CMD2 > $W/pmc4.log 2>&1
python $R/tools/rocpd_summary.py pmc $W/pmc4/pmc4_results.db "$KSUB"; rm -rf $W/pmc4
} > $OUT/${TAG}_pmc.txt
