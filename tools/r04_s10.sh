#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s10}
for v in main nocol0 snop8 slit4 snop16 slit8 vmov8 vlit4 vmov16 main; do
  if [ $v = main ]; then unset BEAGLE_MI355_ENGINE_LIB; else export BEAGLE_MI355_ENGINE_LIB=$ROOT/build/variants/$v/libhmsbeagle-jni.so; fi
  for rep in 1 2; do
  a=$(timeout 200 python bench.py --steps 100 --no-cpu-baseline --no-library-route --no-live-traffic --no-side-records 2>gpurun_out/${TAG}_$v.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us_per_eval'], d['lnL'])")
  echo "$v: A evals/s, kernel us, lnL = $a"
  done
done
