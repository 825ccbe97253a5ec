#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s8}
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
    print('%-28s %9.1f evals/s  %7.4f ms  kernel %7.1f us  other %s' % ('${name}', d['value'], d['ms_per_step'], r['kernel_us_per_eval'], d.get('other_caller')))
except Exception as e:
    print('${name} FAILED', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
run A_s200 A=1 -- --steps 200
run A_always A=1 -- --rescaling always --steps 100
run shard A=1 -- --patterns 12500
run p25k A=1 -- --patterns 25000
run D A=1 -- --config D
run E A=1 -- --config E
run A_ldspad15872 BEAGLE_MI355_WALK_LDS_PAD=15872 -- --steps 60
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/${TAG}_A_line.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_A_line.json').read().strip().splitlines()[-1]); print('A', d['value'], d['ms_per_step'], 'lib', d.get('library_route'), 'shard_point', d.get('shard_point'))
PY
