#!/bin/bash
# Winograd kernel ablations (tools/conv_bench built with -DMD2_ABLATE): 1 = no patch loads / commits after chunk 0, 2 = no U refills, 4 = no MFMAs, 8 = no per-chunk barrier
# (8 only together with 1: the stages are then static).  WMODE picks the kernel (2 / 3 / 4)
B=./tools/conv_bench
for shape in "96 32 104 512 64" "96 16 52 128 128"; do
  for ab in 0 3 11 7 15; do WINO=1 WMODE=${WMODE:-3} $B $shape 3 0 $ab || exit 1; done
done
