#!/bin/bash
# tools/walk_time.py on A/B builds: bash tools/ab_time.sh variant...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for v in "$@"; do
  if [ $v = main ]; then unset BEAGLE_MI355_ENGINE_LIB; else export BEAGLE_MI355_ENGINE_LIB=$ROOT/build/variants/$v/libhmsbeagle-jni.so; fi
  echo "$v: A $(timeout 200 python tools/walk_time.py 2>/dev/null | tail -1) | shard $(timeout 200 python tools/walk_time.py 12500 2>/dev/null | tail -1)"
done
