#!/bin/bash
# one-stream rocprofv3 kernel stats of the headline step (B=32) -> gpurun_out/qs/, printed as a class table + the "other" kernels
R=$PWD; O=$R/gpurun_out/qs; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp && MD2_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/one -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra ${QS_ARGS} > $O/one.log 2>&1 || { tail -5 $O/one.log; exit 1; }
cd $R && python3 tools/profile_summary.py stats $O/one 10 "quick one-stream stats" > $O/one.md && python3 - <<'PY'
import re
t=open("gpurun_out/qs/one.md").read().splitlines()
keep=[l for l in t if l.startswith("| conv") or l.startswith("| loss") or l.startswith("| other") or "sum of kernel" in l]
print("\n".join(keep))
for l in t:
    if l.startswith("| `") and not any(k in l for k in ("conv_", "conv3x3", "stem_", "loss_train")): print(l[:150])
PY
