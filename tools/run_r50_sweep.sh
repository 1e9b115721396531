#!/bin/bash
# ResNet-50 640x192 batch 16: step time under a few launch-heuristic overrides (two streams, the way bench.py measures it)
run() { echo "$1"; env $1 python3 tools/prof_config.py 16 50 3 192 640 f32 10 | grep ms/step; }
run "X=0"; run "X=0"
run "MD2_IGEMM_BIG_TILES=1000"; run "MD2_IGEMM_BIG_TILES=100000"
run "MD2_WGRAD_BLOCKS=768"; run "MD2_WGRAD_BLOCKS=1024"; run "MD2_WGRAD_MAXT=64"
run "MD2_WINO_WGRAD_BLOCKS=768"; run "MD2_NO_WGRAD_ONE=1"; run "MD2_WINO_NH=2"
