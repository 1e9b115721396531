#!/bin/bash
# conv_wino3_kernel (fp32 MFMA, 32-tile workgroups) vs conv_wino3s_kernel (split mode: three bf16 MFMAs per product group), production build of tools/conv_bench.cpp (N H W Cin Cout)
for shape in "96 32 104 64 64" "96 16 52 128 128" "96 8 26 256 256" "96 4 13 512 512" "32 8 26 512 256" "32 16 52 256 128" "32 16 52 128 64" "32 32 104 128 64" "32 32 104 64 32" "32 64 208 96 32" "32 64 208 32 96"; do
  MD2_WINO_MODE=1 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/fp32  /" || exit 1
  MD2_WINO_MODE=4 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/split /" || exit 1
  MD2_WINO_MODE=5 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/split3 /" || exit 1
done
