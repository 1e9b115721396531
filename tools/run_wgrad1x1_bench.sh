#!/bin/bash
# 1x1 / stride-1 weight gradients of the ResNet-50 bottlenecks at 640x192, batch 16 (48 frames): dense loader (default) vs the general per-tap loader (N H W Cin Cout k)
for shape in "48 48 160 64 256" "48 48 160 256 64" "48 48 160 64 64" "48 24 80 128 512" "48 24 80 512 128" "48 12 40 256 1024" "48 12 40 1024 256" "48 6 20 512 2048" "48 6 20 2048 512"; do
  WGRAD=1 ./tools/conv_bench $shape 1 || exit 1
  MD2_NO_WGRAD_ONE=1 WGRAD=1 ./tools/conv_bench $shape 1 || exit 1
done
