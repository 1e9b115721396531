#!/bin/bash
# tools/run_wgrad_blocks_ab.sh <tuning.so>: MD2_WINO_WGRAD_BLOCKS 512 vs 256 vs 320, alternated, three rounds
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so; cp "$1" monodepth2.jl_amd/libmd2hip.so
for round in 1 2 3; do for wb in 512 256 320; do
  for cfg in "4 18 1 128 416" "32 18 1 128 416" "16 50 3 192 640"; do
    MD2_WINO_WGRAD_BLOCKS=$wb python3 tools/prof_config.py $cfg f32 30 2>&1 | grep "ms/step" | sed "s/^/blocks $wb: /"
  done
done; done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
