#!/bin/bash
# usage: tools/disasm_kernel.sh <object or .so> <mangled-name substring> [out.s]   -- gfx950 ISA of one kernel (llvm-objdump), written to out.s (default /tmp/kernel.s)
set -e
src=$(readlink -f "$1"); pat="$2"; out="${3:-/tmp/kernel.s}"
tmp=$(mktemp -d /tmp/md2dis_XXXX); cp "$src" "$tmp/in.o"
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading "$tmp/in.o" > /dev/null
for co in "$tmp"/in.o.*gfx950*; do
  /opt/rocm/lib/llvm/bin/llvm-objdump -d "$co" | awk -v pat="$pat" '/^[0-9a-f]+ <.*>:/{f = index($0, pat) > 0} f{print}' >> "$tmp/all.s"
done
mv "$tmp/all.s" "$out"; rm -rf "$tmp"; wc -l "$out"
