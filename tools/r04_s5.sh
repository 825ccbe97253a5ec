#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s5}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/${TAG}_pytest.log)"
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_${name}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-28s %9.1f evals/s  %7.4f ms  kernel %7.1f us  stored %s lnL %.6f' % ('${name}', d['value'], d['ms_per_step'], r['kernel_us_per_eval'], r.get('per_eval',{}).get('stored'), d['lnL']))
except Exception as e:
    print('${name} FAILED', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
run A_always A=1 -- --rescaling always --steps 100
run A_always_nofusion BEAGLE_MI355_NO_WALK_FUSION=1 -- --rescaling always --steps 100
run shard_always A=1 -- --rescaling always --patterns 12500
run E_default A=1 -- --config E
for c in 56 72 96; do for t in 0 16; do for sim in 1 0; do run shard_c${c}_t${t}_sim$sim BEAGLE_MI355_CHUNK=$c BEAGLE_MI355_CHUNK_TOP=$t BEAGLE_MI355_SCHED_SIM=$sim -- --patterns 12500; done; done; done
run shard_c72_t16_v24 BEAGLE_MI355_CHUNK=72 BEAGLE_MI355_CHUNK_TOP=16 BEAGLE_MI355_VSTEPS=24 -- --patterns 12500
for t in 0 16; do for sim in 1 0; do run p25k_t${t}_sim$sim BEAGLE_MI355_CHUNK_TOP=$t BEAGLE_MI355_SCHED_SIM=$sim -- --patterns 25000; done; done
run p25k_c140_t16 BEAGLE_MI355_CHUNK=140 BEAGLE_MI355_CHUNK_TOP=16 -- --patterns 25000
for t in 0 16; do for sim in 1 0; do run A_t${t}_sim$sim BEAGLE_MI355_CHUNK_TOP=$t BEAGLE_MI355_SCHED_SIM=$sim -- --steps 100; done; done
run D_t16 BEAGLE_MI355_CHUNK_TOP=16 -- --config D
run E_t16 BEAGLE_MI355_CHUNK_TOP=16 -- --config E
BEAGLE_MI355_CHUNK_TOP=16 BEAGLE_MI355_SCHED_SIM=0 BEAGLE_MI355_DUMP_PLAN=1 timeout 200 python bench.py --patterns 25000 --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>&1 >/dev/null | grep "plan:" | sort | uniq -c | head -3
