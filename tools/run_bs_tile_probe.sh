#!/bin/bash
# bf16-tensor implicit GEMM: 64 x 64 against 128 x 64 workgroup tiles inside the configs[4] step (MD2_IGEMM_BIG_TILES; needs a library built with -DMD2_TUNE_ENV): tools/run_bs_tile_probe.sh <tuning.so>
set -e
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so; cp "$1" monodepth2.jl_amd/libmd2hip.so
for big in 1099511627776 1; do
  echo "== MD2_IGEMM_BIG_TILES=$big"
  MD2_IGEMM_BIG_TILES=$big AB_ARMS="1:-1" python3 tools/ab_bench.py 8,18,3,320,1024,bf16_storage 16,50,3,192,640,bf16_storage 32,18,1,128,416,bf16_storage 2>&1 | grep -v amdgpu.ids
done
MD2_IGEMM_BIG_TILES=1 python3 tools/layer_profile.py 8 gpurun_out/r5m/layers_cfg5_big.csv 18 3 320 1024 bf16_storage 2>&1 | grep -v amdgpu.ids | tail -3
python3 tools/layer_profile.py 8 gpurun_out/r5m/layers_cfg5.csv 18 3 320 1024 bf16_storage 2>&1 | grep -v amdgpu.ids | tail -3
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
