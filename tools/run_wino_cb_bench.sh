#!/bin/bash
# fp32-MFMA Winograd kernel: 32 tiles x 32 channels (CB1=1, three workgroups per CU) vs 32 x 64 (two per CU), production build of tools/conv_bench.cpp (N H W Cin Cout)
for shape in "96 32 104 64 64" "32 32 104 32 64" "32 32 104 128 64" "96 16 52 128 128" "96 8 26 256 256" "32 64 208 32 96"; do
  CB1=1 MD2_WINO_PIECES=0 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/32x32      /" || exit 1
  MD2_WINO_PIECES=0 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/32x64      /" || exit 1
  CB1=1 XF=1 MD2_WINO_PIECES=0 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/32x32 xf=1 /" || exit 1
  XF=1 MD2_WINO_PIECES=0 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/32x64 xf=1 /" || exit 1
done
