#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s12}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -1)"; grep -E "^FAILED|Error" gpurun_out/${TAG}_pytest.log | head -5
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records --steps 20 --warmup 5 --step-times > gpurun_out/${TAG}_A20_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_A20_$rep.json').read().strip().splitlines()[-1])
print('A 20/5: %.1f evals/s %.4f ms kernel %.1f us other %s steps %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'], d['other_caller'], d['step_ms']))
PY
done
timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records --steps 200 > gpurun_out/${TAG}_A200.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_A200.json').read().strip().splitlines()[-1])
print('A 200: %.1f evals/s %.4f ms kernel %.1f us other %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'], d['other_caller']))
PY
for cfg in A B C; do
  extra=""; [ $cfg = A ] && extra="--patterns 20000"
  timeout 600 python tools/gradient_bench.py --config $cfg $extra --steps 5 > gpurun_out/${TAG}_grad_$cfg.json 2> gpurun_out/${TAG}_grad_$cfg.err; tail -1 gpurun_out/${TAG}_grad_$cfg.json | cut -c1-420
done
timeout 600 python tools/gradient_bench.py --config A --steps 5 > gpurun_out/${TAG}_grad_A1e5.json 2> gpurun_out/${TAG}_grad_A1e5.err; tail -1 gpurun_out/${TAG}_grad_A1e5.json | cut -c1-420
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_A_line.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_A_line.json').read().strip().splitlines()[-1]); print('A line', d['value'], d['ms_per_step'], 'lib', d.get('library_route'), 'shard_point', d.get('shard_point'), 'partial', d.get('partial_update'), 'cpu', d.get('cpu_baseline'))
PY
