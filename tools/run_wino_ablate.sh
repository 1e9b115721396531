#!/bin/bash
B=./tools/conv_bench
for shape in "96 32 104 512 64" "96 16 52 128 128"; do
  for ab in 0 32 64 3 35 67; do WINO=1 WMODE=${WMODE:-4} $B $shape 3 0 $ab || exit 1; done
done
