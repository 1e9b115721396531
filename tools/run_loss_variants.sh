#!/bin/bash
# tools/run_loss_variants.sh <a.so> [<b.so> ...]: the loss half alone (tools/loss_time.py) at the headline shape (B=32, C=1) and at configuration 5's (B=8, C=3, 1024x320) with the
# tree's library and with each alternative build put in its place, two rounds; then the loss parity tests with each alternative; restores the tree's library.
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so
for round in 1 2; do
  for lib in /tmp/libmd2hip_tree.so "$@"; do
    cp "$lib" monodepth2.jl_amd/libmd2hip.so
    echo "== $lib"; python3 tools/loss_time.py 32 1 2>&1 | grep loss_fwd_bwd; python3 tools/loss_time.py 8 3 320 1024 2>&1 | grep loss_fwd_bwd
  done
done
for lib in "$@"; do
  cp "$lib" monodepth2.jl_amd/libmd2hip.so
  echo "== parity with $lib"; python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x -k "loss or fused_warp or automask or simple_depth or golden or other_shapes" 2>&1 | tail -3
done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
