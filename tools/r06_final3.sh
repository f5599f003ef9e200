#!/bin/bash
# GPU box: the full GPU suite at the final HEAD (tests added after tools/r06_final.sh ran), with the slowest tests listed
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
SECONDS=0
timeout 1100 python -m pytest tests -m gpu -q --durations=25 > $O/r06_final3_gputest.log 2>&1
echo "pytest rc $? after $SECONDS s"; tail -32 $O/r06_final3_gputest.log
timeout 200 python -m pytest tests/test_sashimi_training_gpu.py -m gpu -q -s -k stacked 2>&1 | grep -E "Cauchy launches|stacked vs|passed|failed"
