#!/bin/bash
# tools/run_nan_probe.sh <library.so>: repeated steps of small shapes on bf16 tensors (non-finite gradient arrays, run-to-run differences in bit-reproducible mode), loader on / off
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so; cp "$1" monodepth2.jl_amd/libmd2hip.so
for dl in 1 0; do for shape in "5 3 64 64" "2 1 64 96" "2 1 128 416"; do
echo "== bf16_storage MD2_BS_DL=$dl shape $shape"; NANPROBE_PRECISION=bf16_storage MD2_BS_DL=$dl python3 tools/nan_probe.py $shape 12 2>&1 | grep -E "deterministic=" | tail -12
done; done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
