#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s13}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -1)"; grep -E "^FAILED|Error" gpurun_out/${TAG}_pytest.log | head -5
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_${name}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-28s %9.1f evals/s  %7.4f ms  kernel %7.1f us other %s' % ('${name}', d['value'], d['ms_per_step'], r['kernel_us_per_eval'], d.get('other_caller')))
except Exception as e:
    print('${name} FAILED', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
run shard A=1 -- --patterns 12500
run shard_nofuse BEAGLE_MI355_NO_LAUNCH_FUSION=1 -- --patterns 12500
run shard_b A=1 -- --patterns 12500
run D A=1 -- --config D
run D_nofuse BEAGLE_MI355_NO_LAUNCH_FUSION=1 -- --config D
run E A=1 -- --config E
run A A=1 -- --steps 100
BTL_TIMING=1 BEAGLE_MI355_HOST_TIMING=1 timeout 200 python tools/step_profile.py 12500 2>&1 | tail -16
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_trace" -o kt -- \
   python "$ROOT/bench.py" --patterns 12500 --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > "$ROOT/gpurun_out/${TAG}_trace.json" 2> "$ROOT/gpurun_out/${TAG}_trace.err"; echo "trace rc=$?")
find gpurun_out/${TAG}_trace -name "*.db" -delete 2>/dev/null
python tools/timeline.py gpurun_out/${TAG}_trace 2>&1 | tail -12
