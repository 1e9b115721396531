#!/bin/bash
# batch 4 (the reference's configuration) and batch 32: direct vs Winograd at several workgroup-count thresholds, in-model (per-launch HIP events, one stream)
export MD2_ONE_STREAM=1
for mb in 100000 160 96 48 24; do echo "B=4 min_blocks $mb"; MD2_WINO_MIN_BLOCKS=$mb python3 tools/layer_profile.py 4 gpurun_out/layers_b4_$mb.csv | grep ms; done
for mb in 160 96 48; do echo "B=32 min_blocks $mb"; MD2_WINO_MIN_BLOCKS=$mb python3 tools/layer_profile.py 32 gpurun_out/layers_b32_mb$mb.csv | grep ms; done
