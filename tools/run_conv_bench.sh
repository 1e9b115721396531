#!/bin/bash
export WGRAD=1
for b in 512 768 1152 1536 2304 3072 4608; do echo blocks=$b; MD2_WGRAD_BLOCKS=$b ./tools/conv_bench 96 16 52 128 128 3 0 0; MD2_WGRAD_BLOCKS=$b ./tools/conv_bench 96 8 26 256 256 3 0 0; MD2_WGRAD_BLOCKS=$b ./tools/conv_bench 96 32 104 64 64 3 0 0; MD2_WGRAD_BLOCKS=$b ./tools/conv_bench 96 4 13 512 512 3 0 0; done
