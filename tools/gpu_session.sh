#!/bin/bash
# One GPU-box call of a development session:  bash tools/gpu_session.sh <tag> [what...]
# what: tests | benchA | driverA | shard | trace | host | grad | gradbig     (default: tests benchA shard)
TAG=${1:-s}; shift
WHAT=${*:-tests benchA shard}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out
for w in $WHAT; do case $w in
tests)
  timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/${TAG}_pytest.log)"; grep -E "^FAILED|^ERROR" gpurun_out/${TAG}_pytest.log | head -20;;
testsx)
  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/${TAG}_pytest.log)";;
benchA)
  timeout 400 python bench.py > gpurun_out/${TAG}_bench_A.json 2> gpurun_out/${TAG}_bench_A.err; echo "benchA rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_A.json').read().strip().splitlines()[-1])
print('A', d['value'], 'evals/s median ms', d.get('ms_per_step_median'), 'kernel us', d['roofline']['kernel_us_per_eval'], 'frac', d['roofline']['frac'], 'scale_reads', d['roofline']['per_eval']['scale_reads'], 'lib', d.get('library_route') and d['library_route']['value'], 'shard', d.get('shard_point') and (d['shard_point']['ms_per_step'], d['shard_point']['ms_per_step_median']), 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
PY
  ;;
driverA)
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_A_driver.json 2> gpurun_out/${TAG}_bench_A_driver.err; echo "driverA rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_A_driver.json').read().strip().splitlines()[-1])
print('A(driver cmdline)', d['value'], 'evals/s ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'kernel us', d['roofline']['kernel_us_per_eval'], 'traffic', d['roofline']['traffic'], 'shard', d.get('shard_point') and (d['shard_point']['ms_per_step'], d['shard_point']['ms_per_step_median']), 'move us', d.get('partial_update') and d['partial_update'].get('us_per_branch_move'))
PY
  ;;
shard)
  timeout 300 python bench.py --patterns 12500 --no-cpu-baseline --no-live-traffic > gpurun_out/${TAG}_bench_shard.json 2> gpurun_out/${TAG}_bench_shard.err; echo "shard rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_shard.json').read().strip().splitlines()[-1])
print('shard12500', d['value'], 'evals/s ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'kernel us', d['roofline']['kernel_us_per_eval'], 'lib', d.get('library_route') and d['library_route']['value'])
PY
  ;;
trace)
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_trace" -o kt -- \
     python "$ROOT/bench.py" --patterns 12500 --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > "$ROOT/gpurun_out/${TAG}_trace.json" 2> "$ROOT/gpurun_out/${TAG}_trace.err"; echo "trace rc=$?")
  find gpurun_out/${TAG}_trace -name "*.db" -delete 2>/dev/null
  python tools/timeline.py gpurun_out/${TAG}_trace 2>&1 | tail -40;;
host)
  BEAGLE_MI355_HOST_TIMING=1 timeout 200 python tools/step_profile.py 12500 2>&1 | tail -14;;
grad)
  timeout 300 python tools/gradient_bench.py --patterns 100000 > gpurun_out/${TAG}_gradient_1e5.json 2> gpurun_out/${TAG}_gradient_1e5.err; echo "grad rc=$?"; tail -3 gpurun_out/${TAG}_gradient_1e5.json | cut -c1-600
  timeout 300 python tools/gradient_bench.py > gpurun_out/${TAG}_gradient.json 2> gpurun_out/${TAG}_gradient.err; echo "grad20k rc=$?"; tail -3 gpurun_out/${TAG}_gradient.json | cut -c1-600;;
esac; done
