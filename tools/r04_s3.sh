#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s3}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/${TAG}_pytest.log)"
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_${name}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-28s %9.1f evals/s  %7.4f ms  kernel %7.1f us  launches %s lnL %.6f' % ('${name}', d['value'], d['ms_per_step'], r['kernel_us_per_eval'], r.get('launches_per_eval'), d['lnL']))
except Exception as e:
    print('${name} FAILED', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
run shard_default A=1 -- --patterns 12500
run shard_nofusion BEAGLE_MI355_NO_WALK_FUSION=1 -- --patterns 12500
run shard_copyengine BEAGLE_MI355_COPY_ENGINE_UPLOADS=1 -- --patterns 12500
for c in 16 24 32 56; do run shard_chunk$c BEAGLE_MI355_CHUNK=$c -- --patterns 12500; done
run A_default A=1 -- --steps 100
run A_nofusion BEAGLE_MI355_NO_WALK_FUSION=1 -- --steps 100
for c in 60 100; do run A_chunk$c BEAGLE_MI355_CHUNK=$c -- --steps 100; done
run p25k_default A=1 -- --patterns 25000
run p25k_nofusion BEAGLE_MI355_NO_WALK_FUSION=1 -- --patterns 25000
run p50k_default A=1 -- --patterns 50000
run p50k_nofusion BEAGLE_MI355_NO_WALK_FUSION=1 -- --patterns 50000
run E_default A=1 -- --config E
run E_nofusion BEAGLE_MI355_NO_WALK_FUSION=1 -- --config E
run D_default A=1 -- --config D
run D_nofusion BEAGLE_MI355_NO_WALK_FUSION=1 -- --config D
run shard_forcesharded A=1 -- --patterns 12500 --force-sharded
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/${TAG}_A_line.json 2> gpurun_out/${TAG}_A_line.err; echo "A line rc=$?"; python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_A_line.json').read().strip().splitlines()[-1])
print('A', d['value'], d['ms_per_step'], 'shard_point', d.get('shard_point'), 'partial', d.get('partial_update'), 'lib', d.get('library_route'))
PY
BTL_TIMING=1 BEAGLE_MI355_HOST_TIMING=1 timeout 200 python tools/step_profile.py 12500 2>&1 | tail -18
