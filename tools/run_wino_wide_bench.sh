#!/bin/bash
# conv_wino3s_kernel (32 tiles x 32 channels, 256 threads) vs conv_wino3w_kernel (32 tiles x 64 channels, 512 threads), exact three-piece split, production build of tools/conv_bench.cpp (N H W Cin Cout)
for shape in "96 32 104 64 64" "96 16 52 128 128" "96 8 26 256 256" "96 4 13 512 512" "32 8 26 512 256" "32 16 52 256 128" "32 16 52 128 64" "32 32 104 128 64" "32 32 104 64 64"; do
  MD2_WINO_MODE=5 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/narrow      /" || exit 1
  WIDE=1 MD2_WINO_MODE=5 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/wide        /" || exit 1
  XF=1 MD2_WINO_MODE=5 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/narrow xf=1 /" || exit 1
  WIDE=1 XF=1 MD2_WINO_MODE=5 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/wide   xf=1 /" || exit 1
done
