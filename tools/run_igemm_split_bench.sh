#!/bin/bash
# direct implicit GEMM, fp32 MFMA vs exact three-piece split on the bf16 pipe (production build of tools/conv_bench.cpp): N H W Cin Cout k [STRIDE]
run() { local st=$1; shift; STRIDE=$st ./tools/conv_bench_prod "$@" | sed "s/^/fp32   s$st /"; SPLIT3=1 STRIDE=$st ./tools/conv_bench_prod "$@" | sed "s/^/split3 s$st /"; }
# ResNet-18 at 416x128, batch 32: the three down-sampling 3x3 convolutions and their 1x1 shortcuts, the pose squeezer
run 2 96 32 104 64 128 3; run 2 96 16 52 128 256 3; run 2 96 8 26 256 512 3
run 2 96 32 104 64 128 1; run 2 96 8 26 256 512 1; run 1 96 4 13 512 256 1
# ResNet-50 at 640x192, batch 16 (48 frames): bottleneck 1x1 convolutions of the four stages
run 1 48 48 160 64 256 1; run 1 48 48 160 256 64 1; run 1 48 24 80 128 512 1; run 1 48 24 80 512 128 1
run 1 48 12 40 256 1024 1; run 1 48 12 40 1024 256 1; run 1 48 6 20 512 2048 1; run 1 48 6 20 2048 512 1
run 2 48 48 160 256 512 1; run 2 48 24 80 128 128 3
