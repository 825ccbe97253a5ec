#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s18}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -1)"; grep -E "^FAILED|Error" gpurun_out/${TAG}_pytest.log | head -5
for cfg in B C; do
  for two in 0 1; do
    echo "$cfg two_pass=$two: $(BEAGLE_MI355_PRE_TWO_PASS=$two timeout 600 python tools/gradient_bench.py --config $cfg --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_gradient'], 'ms per gradient; likelihood', d['ms_per_likelihood_same_driver'], 'grad_norm', d['grad_norm'], 'lnL', d['lnL'])")"
  done
done
timeout 300 python bench.py --config E --no-cpu-baseline --no-live-traffic --no-library-route 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('E', d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'], d['lnL'])"
