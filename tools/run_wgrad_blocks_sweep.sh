#!/bin/bash
# tools/run_wgrad_blocks_sweep.sh <tuning.so>: step time against the block-count targets of the weight-gradient launches (split-K width: MD2_WINO_WGRAD_BLOCKS, default 512;
# MD2_WGRAD_BLOCKS, default by shape) at batch 4 and 32 -- do narrower weight gradients, which leave CUs to the dependency chain they overlap, pay at batch 4?
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so; cp "$1" monodepth2.jl_amd/libmd2hip.so
for cfg in "4 18 1 128 416" "32 18 1 128 416"; do
  for wb in 512 384 256 192 128 64; do
    MD2_WINO_WGRAD_BLOCKS=$wb python3 tools/prof_config.py $cfg f32 30 2>&1 | grep "ms/step" | sed "s/^/wino wgrad blocks $wb: /"
  done
  for tb in 64 128 256; do
    MD2_WGRAD_BLOCKS=$tb python3 tools/prof_config.py $cfg f32 30 2>&1 | grep "ms/step" | sed "s/^/tap wgrad blocks $tb: /"
  done
done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
