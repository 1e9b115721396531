#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <other.so> [ab_bench.py args]   (runs tools/ab_bench.py with the tree's library, then with <other.so> in its place, then restores)
set -e
other="$1"; shift
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so
for round in 1 2; do
  echo "== tree library"; python3 tools/ab_bench.py "$@" 2>&1 | grep -v amdgpu.ids
  cp "$other" monodepth2.jl_amd/libmd2hip.so
  echo "== $other"; python3 tools/ab_bench.py "$@" 2>&1 | grep -v amdgpu.ids
  cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
done
