#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s16}
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_${name}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-28s %9.1f evals/s  %7.4f ms  kernel %7.1f us stored %s' % ('${name}', d['value'], d['ms_per_step'], r['kernel_us_per_eval'], r.get('per_eval',{}).get('stored')))
except Exception as e:
    print('${name} FAILED', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
for rep in 1 2; do
for c in 150 200 256 350 500; do run A_c${c}_$rep BEAGLE_MI355_CHUNK=$c -- --steps 60; done
done
for c in 150 256; do run p50k_c$c BEAGLE_MI355_CHUNK=$c -- --patterns 50000; done
