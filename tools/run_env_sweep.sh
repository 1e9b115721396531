#!/bin/bash
# step time under a few environment settings (two streams): tools/run_env_sweep.sh [-b BATCH] "VAR=val VAR2=val" ...
B=32; if [ "$1" = "-b" ]; then B=$2; shift 2; fi
for e in "" "$@"; do
  echo "== ${e:-default}"
  env $e python3 tools/prof_config.py $B 18 1 128 416 f32 30 | tail -1
done
