#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s9}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -1)"
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_${name}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-28s %9.1f evals/s  %7.4f ms  kernel %7.1f us' % ('${name}', d['value'], d['ms_per_step'], r['kernel_us_per_eval']))
except Exception as e:
    print('${name} FAILED', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
run A_s200 A=1 -- --steps 200
run A_w5_s20 A=1 -- --steps 20 --warmup 5
run A_w60_s20 A=1 -- --steps 20 --warmup 60
run A_w5_s20_b A=1 -- --steps 20 --warmup 5
run A_w60_s20_b A=1 -- --steps 20 --warmup 60
run A_always A=1 -- --rescaling always --steps 100
run shard A=1 -- --patterns 12500
run p25k A=1 -- --patterns 25000
run D A=1 -- --config D
run E A=1 -- --config E
