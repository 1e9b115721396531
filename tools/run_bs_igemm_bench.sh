#!/bin/bash
# bf16-tensor implicit GEMM (md2_config.precision = 2) on the configs[4] encoder shapes: 64 x 64 against 128 x 64 tiles, register-staged against direct-to-LDS loads
# (MD2_BS_DL), and the ablations of the K loop (1 = no global loads, 2 = no LDS traffic, 4 = no MFMAs, 8 = no barrier, 16 = no epilogue stores)
for shape in "24 80 256 64 64" "24 40 128 128 128" "24 20 64 256 256" "24 10 32 512 512"; do
  for tile in 3 2; do
    for ab in 0 1 2 4 8 16; do BF16S=1 ./tools/conv_bench $shape 3 $tile $ab; done
    BF16S=1 MD2_BS_DL=1 ./tools/conv_bench $shape 3 $tile 0 | sed 's/$/  [direct-to-LDS]/'
  done
done
