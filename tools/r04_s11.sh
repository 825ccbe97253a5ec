#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s11}
for rep in 1 2; do
for sp in -1 0.05 0.3 1.0; do
  timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records --steps 20 --warmup 5 --spinup $sp --step-times > gpurun_out/${TAG}_sp$sp.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_sp$sp.json').read().strip().splitlines()[-1])
print('spinup $sp: %.1f evals/s %.4f ms kernel %.1f us spin_evals %d steps %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'], d['spinup_evaluations'], d['step_ms']))
PY
done; done
timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records --steps 150 --warmup 5 --step-times > gpurun_out/${TAG}_series.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_series.json').read().strip().splitlines()[-1])
print('series', d['step_ms'])
PY
for cfg in A B C; do
  extra=""; [ $cfg = A ] && extra="--patterns 20000"
  timeout 600 python tools/gradient_bench.py --config $cfg $extra --steps 5 > gpurun_out/${TAG}_grad_$cfg.json 2> gpurun_out/${TAG}_grad_$cfg.err; tail -1 gpurun_out/${TAG}_grad_$cfg.json | cut -c1-600
done
timeout 600 python tools/gradient_bench.py --config A --steps 5 > gpurun_out/${TAG}_grad_A1e5.json 2> gpurun_out/${TAG}_grad_A1e5.err; tail -1 gpurun_out/${TAG}_grad_A1e5.json | cut -c1-600
