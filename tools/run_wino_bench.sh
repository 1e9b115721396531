#!/bin/bash
# direct implicit GEMM vs the two fused Winograd kernels on the 3x3 stride-1 shapes of the step at batch 32 (N H W Cin Cout k)
for shape in "96 32 104 64 64" "96 16 52 128 128" "96 8 26 256 256" "96 4 13 512 512" "32 8 26 512 256" "32 16 52 256 128" "32 32 104 128 64" "96 32 104 512 64"; do
  ./tools/conv_bench $shape 3 || exit 1
  WINO=1 WMODE=2 ./tools/conv_bench $shape 3 || exit 1
  WINO=1 WMODE=3 ./tools/conv_bench $shape 3 || exit 1
  WINO=1 WMODE=3 XF=1 ./tools/conv_bench $shape 3 || exit 1
done
