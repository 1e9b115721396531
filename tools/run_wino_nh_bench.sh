#!/bin/bash
# conv_wino3_kernel with 64-tile (MD2_WINO_NH=2) vs 32-tile (MD2_WINO_NH=1) workgroups on the 3x3 stride-1 shapes of the step at batch 32, production build of tools/conv_bench.cpp
# (N H W Cin Cout): encoder stages, then the decoder layers and their input gradients
for shape in "96 32 104 64 64" "96 16 52 128 128" "96 8 26 256 256" "96 4 13 512 512" "32 8 26 512 256" "32 8 26 256 512" "32 16 52 256 128" "32 16 52 128 256" "32 16 52 128 64" "32 16 52 64 128" "32 32 104 128 64" "32 32 104 64 128" "32 32 104 64 32" "32 32 104 32 64" "32 64 208 96 32" "32 64 208 32 96" "64 4 13 512 256" "64 4 13 256 256"; do
  for nh in 2 1; do MD2_WINO_NH=$nh WINO=1 WMODE=3 ./tools/conv_bench_prod $shape 3 | sed "s/^/nh=$nh /" || exit 1; done
done
