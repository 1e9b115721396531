#!/bin/bash
# same-box probe of tuning switches (md2_tune / md2_tune_set names, read from the environment by a tuning build of the library:
#   make -C monodepth2.jl_amd/csrc FLAGS+=-DMD2_TUNE_ENV in a copy of the tree -> tools/ab/libmd2hip_tune.so), arms alternated twice.
# usage: tools/env_probe.sh "X=0|MD2_WGRAD_TAIL_CHAIN=2|MD2_WGRAD_TAIL_CHAIN=4 MD2_STEM_WGRAD_FORK=1" [ab_bench.py configs]     (arms separated by |)
set -e
arms="$1"; shift
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so
cp tools/ab/libmd2hip_tune.so monodepth2.jl_amd/libmd2hip.so
IFS='|' read -ra A <<< "$arms"
for round in 1 2; do
  for a in "${A[@]}"; do
    echo "== $a"; env $a AB_ARMS=1:-1 python3 tools/ab_bench.py "$@" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
