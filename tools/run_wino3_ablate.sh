#!/bin/bash
# conv_wino3_kernel ablation matrix (tools/conv_bench built with -DMD2_ABLATE): bit 1 = no patch loads / commits after chunk 0, 2 = no U refills, 4 = no MFMAs,
# 8 = no per-chunk barrier, 16 = conflict-free (wrong) LDS read addresses, 32 = no LDS reads / input transform
B=./tools/conv_bench
for shape in "96 32 104 64 64" "96 16 52 128 128" "96 8 26 256 256"; do
  for ab in 0 4 32 35 43 3 11 1 2 16 36 39; do WINO=1 WMODE=3 $B $shape 3 0 $ab || exit 1; done
done
