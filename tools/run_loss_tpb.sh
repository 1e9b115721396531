#!/bin/bash
# tiles per workgroup of loss_train_kernel (MD2_LOSS_TPB; needs a library built with -DMD2_TUNE_ENV in place of the product one): tools/run_loss_tpb.sh <tuning.so>
set -e
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so; cp "$1" monodepth2.jl_amd/libmd2hip.so
for tpb in 1 2 3 4 6; do
  MD2_LOSS_TPB=$tpb python3 tools/loss_time.py 32 1 128 416 2>&1 | grep loss_fwd_bwd | sed "s/^/tpb=$tpb /"
  MD2_LOSS_TPB=$tpb python3 tools/loss_time.py 8 3 320 1024 2>&1 | grep loss_fwd_bwd | sed "s/^/tpb=$tpb /"
done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
