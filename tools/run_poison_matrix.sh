#!/bin/bash
# tools/run_poison_matrix.sh: the LDS-poisoning probe over encoder depths, product forms, Winograd modes and shapes (each line: non-finite / differing steps must be 0)
run() { env "$@" 2>&1 | grep -v amdgpu.ids | grep -E "poison|non-finite arrays|Error|error" | cut -c1-230; }
run PP_DEPTH=50 python3 tools/poison_probe.py 2 3 64 96 f32 3
run PP_DEPTH=50 python3 tools/poison_probe.py 2 3 64 96 bf16_storage 3
run PP_PRODUCTS=2,2 python3 tools/poison_probe.py 5 3 64 64 f32 3
run PP_PRODUCTS=0,0 python3 tools/poison_probe.py 5 3 64 64 f32 3
run PP_PRODUCTS=3,3 python3 tools/poison_probe.py 5 3 64 64 f32 3
run PP_WINO=3,0 python3 tools/poison_probe.py 5 3 64 64 f32 3
run PP_WINO=2,0 python3 tools/poison_probe.py 2 1 128 416 f32 3
run PP_WINO=11,0 python3 tools/poison_probe.py 2 1 128 416 f32 3
run PP_WINO=0,0 python3 tools/poison_probe.py 3 1 96 160 f32 3
run X=1 python3 tools/poison_probe.py 3 1 96 160 f32 3
run X=1 python3 tools/poison_probe.py 1 3 192 640 f32 3
run X=1 python3 tools/poison_probe.py 1 3 192 640 bf16_storage 3
run X=1 python3 tools/poison_probe.py 3 1 96 160 bf16 3
