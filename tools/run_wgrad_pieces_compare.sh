#!/bin/bash
# in-model per-launch HIP-event tables (one stream) of the weight gradients with fp32-MFMA products vs the exact three-piece split: gpurun_out/layers_wg{0,3}_b{32,4}.csv
export MD2_ONE_STREAM=1
for b in 32 4; do
  for p in 0 3; do MD2_WGRAD_PIECES=$p MD2_WINO_PIECES=$p python3 tools/layer_profile.py $b gpurun_out/layers_wg${p}_b${b}.csv || exit 1; done
done
