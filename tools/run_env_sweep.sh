#!/bin/bash
# step time at the headline shape under a few environment settings (two streams): tools/run_env_sweep.sh "VAR=val VAR2=val" ...
for e in "" "$@"; do
  echo "== ${e:-default}"
  env $e python3 tools/prof_config.py 32 18 1 128 416 f32 20 | tail -1
done
