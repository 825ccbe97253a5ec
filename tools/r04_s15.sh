#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s15}
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_${name}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-28s %9.1f evals/s  %7.4f ms  kernel %7.1f us launches %s lnL %.9f' % ('${name}', d['value'], d['ms_per_step'], r['kernel_us_per_eval'], r.get('launches_per_eval'), d['lnL']))
except Exception as e:
    print('${name} FAILED', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
run C_default A=1 -- --config C --steps 40
for k in 1 2 4 8 16; do run C_dfs$k BEAGLE_MI355_SCHED=dfs:$k -- --config C --steps 40; done
run C_asap BEAGLE_MI355_SCHED=asap -- --config C --steps 40
run Balways_default A=1 -- --config B --rescaling always --steps 30
run Balways_dfs4 BEAGLE_MI355_SCHED=dfs:4 -- --config B --rescaling always --steps 30
BEAGLE_MI355_SCHED=dfs:2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "engine_matches_oracle or tree_shapes" 2>&1 | tail -2
