#!/bin/bash
# Winograd-domain weight gradient, fp32 MFMA kernel (64 x 32-channel workgroups) vs exact-split kernel (32 x 32), production build of tools/conv_bench.cpp: N H W Cin Cout
for shape in "96 32 104 64 64" "96 16 52 128 128" "96 8 26 256 256" "96 4 13 512 512" "32 8 26 512 256" "32 16 52 256 128" "32 32 104 128 64"; do
  for p in 0 3; do MD2_WGRAD_PIECES=$p WGRAD=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/pieces=$p /" || exit 1; done
done
