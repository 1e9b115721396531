#!/bin/bash
# all-taps weight-gradient ablations: 1 no global loads, 64 no LDS commit, 32 no LDS operand reads, 4 no MFMA, 16 no atomics tail, 8 no barriers
export WGRAD=1
for shape in "96 32 104 64 64" "96 16 52 128 128" "96 4 13 512 512"; do
  for ab in 0 16 1 65 32 8 4 20; do ./tools/conv_bench $shape 3 0 $ab; done
done
