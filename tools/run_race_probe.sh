#!/bin/bash
# tools/run_race_probe.sh [other.so ...]: tools/race_probe.py on the stress shapes with the tree's library, then with each other build in its place (restored afterwards)
shapes=${RACE_SHAPES:-"4,1,128,416,18 2,3,96,160,50 5,3,64,64,18 2,1,64,96,18"}
run() { for p in f32 bf16_storage; do for s in $shapes; do python3 tools/race_probe.py ${s//,/ } $p ${RACE_REPS:-64} 2>&1 | grep race_probe; done; done; }
echo "== tree library"; run
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so
for other in "$@"; do cp "$other" monodepth2.jl_amd/libmd2hip.so; echo "== $other"; run; done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
