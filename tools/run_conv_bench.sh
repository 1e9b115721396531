#!/bin/bash
for t in 2 3 6; do ./tools/conv_bench 96 16 52 128 128 3 $t 0; done
for t in 2 3; do ./tools/conv_bench 96 8 26 256 256 3 $t 0; done
for t in 2 3; do ./tools/conv_bench 96 32 104 64 64 3 $t 0; done
for t in 2 3; do ./tools/conv_bench 96 4 13 512 512 3 $t 0; done
echo "--- ablations on 128->128 tile 2 (1=no gload 2=no lstore 4=no mfma 8=no barrier)"
for ab in 0 1 3 4; do ./tools/conv_bench 96 16 52 128 128 3 2 $ab; done
echo "--- large M"
for t in 2 3; do ./tools/conv_bench 384 16 52 128 128 3 $t 0; done
