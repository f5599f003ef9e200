#!/bin/bash
# Same-box A/B of the launch batching of the training step (tools/ab/libdws_prev.so = the build before it):
#   weight preparation as job tables (PrepBatch), weight-norm adjoint inside the split-K reduce, LN scalar sums in one launch
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/ab_lib.sh "weight_prep\|pack_a_frag\|fold_weight\|wgrad_reduce\|weight_norm_bwd\|sum_pair\|row_sum" \
   --config unet_d128_n6_T200 --mode train --precision ${1:-f32} --steps 6 --warmup 2 --no-roofline
