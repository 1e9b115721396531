#!/bin/bash
# in-model per-launch HIP-event tables (one stream): direct implicit GEMM vs the automatic Winograd choice vs the pinned kernels: gpurun_out/layers_{direct,auto,w4}.csv
export MD2_ONE_STREAM=1
MD2_NO_WINO=1 python3 tools/layer_profile.py 32 gpurun_out/layers_direct.csv || exit 1
python3 tools/layer_profile.py 32 gpurun_out/layers_auto.csv || exit 1
MD2_WINO_MODE=3 MD2_WINO_MIN_BLOCKS=1 python3 tools/layer_profile.py 32 gpurun_out/layers_w3.csv || exit 1
