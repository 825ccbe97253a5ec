#!/bin/bash
# bench.py (config A and the 12 500-pattern shard) on A/B builds of the engine:  bash tools/ab_variants.sh <tag> [variant ...]
# "main" = the tracked build; others = build/variants/<name>/ (tools/build_variant.sh)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
for v in "$@"; do
  if [ $v = main ]; then unset BEAGLE_MI355_ENGINE_LIB; else export BEAGLE_MI355_ENGINE_LIB=$ROOT/build/variants/$v/libhmsbeagle-jni.so; fi
  for rep in 1 2; do
  a=$(timeout 200 python bench.py --steps 100 --no-cpu-baseline --no-library-route 2>gpurun_out/${TAG}_$v.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us_per_eval'], d['lnL'])")
  s=$(timeout 200 python bench.py --steps 100 --no-cpu-baseline --no-library-route --patterns 12500 2>>gpurun_out/${TAG}_$v.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us_per_eval'])")
  echo "$v: A evals/s, kernel us, lnL = $a | shard12500 = $s"
  done
done
