#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s6}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/${TAG}_pytest.log)"
for v in 8 16 24; do for p in 12500 25000; do
  echo "VSTEPS=$v patterns=$p: $(BEAGLE_MI355_VSTEPS=$v timeout 200 python tools/partial_update_bench.py $p 300 2>&1 | tail -1)"
done; done
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
for v in 8 16 24; do run shard_v$v BEAGLE_MI355_VSTEPS=$v -- --patterns 12500; done
for v in 16 24 32; do run p25k_v$v BEAGLE_MI355_VSTEPS=$v -- --patterns 25000; done
run B_default A=1 -- --config B --steps 30
run C_default A=1 -- --config C --steps 30
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_A_line.json 2> gpurun_out/${TAG}_A_line.err; echo "A line rc=$?"; python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_A_line.json').read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ('value','ms_per_step','shard_point','partial_update','library_route','other_caller')}, indent=1))
print(json.dumps(d['roofline'], indent=1)[:1500]); print(d['cpu_baseline'])
PY
