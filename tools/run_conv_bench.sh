#!/bin/bash
# igemm ablations (1=no gload 2=no lds 4=no mfma 8=no barrier 16=no epilogue stores) on the encoder stage shapes
# (last run: gload costs 10-17%, epilogue 3-5%; 256x64 / 128x128 / 256x128 tiles were 20-40% SLOWER than 128x64: 1-2 waves per SIMD)
for shape in "96 32 104 64 64" "96 16 52 128 128" "96 8 26 256 256" "96 4 13 512 512" "96 32 104 512 64"; do
  for ab in 0 1 16 17 4 8; do ./tools/conv_bench $shape 3 0 $ab; done
done
