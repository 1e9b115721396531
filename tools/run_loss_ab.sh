#!/bin/bash
# tools/run_loss_ab.sh <other.so>: tools/loss_time.py (the loss half alone) at the headline shape (B=32, C=1) and at configuration 5's (B=8, C=3, 1024x320), tree library and <other.so> alternated
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so
for round in 1 2; do
  for lib in /tmp/libmd2hip_tree.so "$1"; do
    cp "$lib" monodepth2.jl_amd/libmd2hip.so
    echo "== $lib"; python3 tools/loss_time.py 32 1 2>&1 | grep loss_fwd_bwd; python3 tools/loss_time.py 8 3 320 1024 2>&1 | grep loss_fwd_bwd
  done
done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
