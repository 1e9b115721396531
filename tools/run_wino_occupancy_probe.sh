#!/bin/bash
# conv_wino3s_kernel (exact three-piece split) at two workgroups per CU (as shipped) and at ONE (dynamic LDS padded by 40 KB): how much of its time is latency that
# more resident waves would cover?  tools/conv_bench_prod = tools/conv_bench.cpp built with -DMD2_TUNE_ENV (tools/README.md): launch_wino3s_t then adds MD2_WINO_LDS_PAD bytes of LDS.
for shape in "96 16 52 128 128" "96 8 26 256 256" "96 4 13 512 512" "32 16 52 256 128" "32 32 104 128 64"; do
  MD2_WINO_LDS_PAD=0 MD2_WINO_MODE=5 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/2 per CU  /" || exit 1
  MD2_WINO_LDS_PAD=40000 MD2_WINO_MODE=5 WINO=1 ./tools/conv_bench_prod $shape 3 | sed "s/^/1 per CU  /" || exit 1
done
