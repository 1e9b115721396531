#!/bin/bash
# tools/run_nan_probe.sh <experimental library.so> "VAR=val ..." ...: repeated steps of the two shapes that failed, one arm per argument
lib="$1"; shift
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so; cp "$lib" monodepth2.jl_amd/libmd2hip.so
for arm in "$@"; do for shape in "5 3 64 64" "2 1 64 96"; do
  echo "== $arm  shape $shape"; env $arm python3 tools/nan_probe.py $shape 12 2>&1 | grep -E "deterministic=" | tail -4
done; done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
