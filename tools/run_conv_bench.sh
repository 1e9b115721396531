#!/bin/bash
export WGRAD=1
echo "--- wgrad ablations (1=no gload 2=no lds 4=no mfma 8=no barrier 16=no atomics 32=no lds reads 64=no lds writes)"
for ab in 0 32 64 2 1 33 65; do ./tools/conv_bench 96 16 52 128 128 3 0 $ab; done
for ab in 0 32 64 2 1; do ./tools/conv_bench 96 32 104 64 64 3 0 $ab; done
