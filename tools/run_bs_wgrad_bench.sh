#!/bin/bash
# all-taps weight gradient on bf16 tensors on the configs[4] shapes: the transposed-read kernel (conv3x3_wgrad_taps_tr_kernel) against the channel-major one (MD2_NO_TR_TAPS=1),
# with ablations (1: global loads for the first tile only, 2: LDS commit for the first tile only, 4: no MFMAs)
for shape in "24 80 256 64 64" "24 40 128 128 128" "24 20 64 256 256" "24 10 32 512 512" "8 160 512 96 32"; do
  MD2_NO_TR_TAPS=1 WGRAD=1 BF16S=1 ./tools/conv_bench $shape 3 0 0 | sed 's/$/  [channel-major tiles]/'
  for ab in 0 1 2 4 7; do WGRAD=1 BF16S=1 ./tools/conv_bench $shape 3 0 $ab; done
done
