#!/bin/bash
# one-stream rocprofv3 kernel stats of other configurations: args = name B depth C H W precision (repeatable groups of 7)
R=$PWD; O=$R/gpurun_out/cfg5; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
while [ $# -ge 7 ]; do
  n=$1; shift
  cd /tmp && MD2_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o t -- python3 $R/tools/prof_config.py $1 $2 $3 $4 $5 $6 5 > $O/$n.log 2>&1 || { tail -5 $O/$n.log; exit 1; }
  cd $R && python3 tools/profile_summary.py stats $O/$n 7 "$n one-stream" > $O/$n.md; grep "ms/step" $O/$n.log
  shift 6
done
