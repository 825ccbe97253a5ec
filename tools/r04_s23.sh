#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
run() { # label, env...
  local label=$1; shift
  echo "$label: $(env "$@" timeout 200 python bench.py --config A --steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'evals/s', d['ms_per_step'], 'ms kernel', r['kernel_us_per_eval'])")"
}
for x in noldswait notopwait novmwait "noldswait,notopwait,novmwait" nofma "nofma,noldswait,notopwait,novmwait"; do
  name=$(echo $x | tr ',' '_')
  bash tools/build_variant.sh $name WALK4_EXPERIMENT=$x > /dev/null 2>&1 || echo "build $name failed"
done
EXTRA=""; run "A depth2 full" X=1
for x in noldswait notopwait novmwait noldswait_notopwait_novmwait nofma nofma_noldswait_notopwait_novmwait; do
  run "A without [$x]" BEAGLE_MI355_ENGINE_LIB=$ROOT/build/variants/$x/libhmsbeagle-jni.so
  run "A without [$x] and stores" BEAGLE_MI355_ENGINE_LIB=$ROOT/build/variants/$x/libhmsbeagle-jni.so BEAGLE_MI355_ABLATE=1
done
run "A depth2 full, no stores" BEAGLE_MI355_ABLATE=1
