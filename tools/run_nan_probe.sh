#!/bin/bash
# tools/run_nan_probe.sh [library.so] "VAR=val ..." ...: repeated steps of the small shapes, one arm per argument (an experimental / tuning library may be put in place first)
if [ -f "$1" ]; then cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so; cp "$1" monodepth2.jl_amd/libmd2hip.so; swapped=1; shift; fi
for arm in "$@"; do for shape in ${NANPROBE_SHAPES:-"5,3,64,64 2,1,64,96 2,1,128,416"}; do
  echo "== $arm  shape $shape"; env $arm python3 tools/nan_probe.py ${shape//,/ } ${NANPROBE_REPS:-12} 2>&1 | grep -E "deterministic=True" | tail -4
done; done
[ -n "$swapped" ] && cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
