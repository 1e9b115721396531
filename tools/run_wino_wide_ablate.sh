#!/bin/bash
# where a launch of conv_wino3w_kernel spends its time (ablation build tools/conv_bench, -DMD2_ABLATE): abl bits 1 no patch loads / commits, 2 no U refills, 4 no MFMAs, 8 no barrier,
# 32 no LDS reads / transform / split, 64 return after the main loop, 128 no prologue loads, 256 plain stores of one accumulator set instead of the output transform + epilogue
set -e
for shape in "96 16 52 128 128" "96 32 104 64 64"; do
  for abl in 0 47 111 175 239 303 431; do
    WIDE=1 MD2_WINO_MODE=5 WINO=1 ./tools/conv_bench $shape 3 0 $abl 2>&1 | tee /tmp/abl.out; if grep -q "Memory access fault" /tmp/abl.out; then exit 1; fi
  done
done
