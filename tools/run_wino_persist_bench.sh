#!/bin/bash
# conv_wino3s_kernel (one item per workgroup) vs conv_wino3p_kernel (persistent, round 6), exact three-piece split; tools/cb_link = tools/conv_bench.cpp linked against the library's objects (N H W Cin Cout)
for shape in "96 16 52 128 128" "96 8 26 256 256" "96 4 13 512 512" "32 8 26 512 256" "32 16 52 256 128" "32 16 52 384 128" "32 32 104 192 64" "32 32 104 128 64" "32 64 208 96 32"; do
  PERSIST=0 MD2_WINO_MODE=5 WINO=1 ./tools/cb_link $shape 3 | sed "s/^/one-item   /" || exit 1
  PERSIST=1 MD2_WINO_MODE=5 WINO=1 ./tools/cb_link $shape 3 | sed "s/^/persistent /" || exit 1
done
