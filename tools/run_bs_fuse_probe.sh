#!/bin/bash
# bf16 tensors: BatchNorm + ReLU folded into the loaders of conv2 / its weight gradient (default) against materialised activations (MD2_NO_BN_FUSE=1; needs a -DMD2_TUNE_ENV library)
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so; cp "$1" monodepth2.jl_amd/libmd2hip.so
for r in 1 2; do
for f in 0 1; do
  echo "== MD2_NO_BN_FUSE=$f"
  if [ $f = 1 ]; then export MD2_NO_BN_FUSE=1; else unset MD2_NO_BN_FUSE; fi
  AB_ARMS="1:-1" python3 tools/ab_bench.py 8,18,3,320,1024,bf16_storage 16,50,3,192,640,bf16_storage 32,18,1,128,416,bf16_storage 4,18,1,128,416,bf16_storage 2>&1 | grep -v amdgpu.ids
done; done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
