#!/bin/bash
# batch 4 (the reference's configuration): direct vs Winograd at several workgroup-count thresholds, in-model
export MD2_ONE_STREAM=1
for mb in 100000 160 96 48 24; do echo "min_blocks $mb"; MD2_WINO_MIN_BLOCKS=$mb python3 tools/layer_profile.py 4 gpurun_out/layers_b4_$mb.csv | grep ms; done
