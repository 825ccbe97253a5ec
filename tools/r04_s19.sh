#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s19}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -1)"; grep -E "^FAILED|Error" gpurun_out/${TAG}_pytest.log | head -5
for cfg in B C; do
  for two in 0 2 1; do
    echo "$cfg edge_two_step=$two: $(BEAGLE_MI355_EDGE_TWO_STEP=$two timeout 600 python tools/gradient_bench.py --config $cfg --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_gradient'], 'ms per gradient; likelihood', d['ms_per_likelihood_same_driver'], 'grad_norm', d['grad_norm'], 'lnL', d['lnL'])")"
  done
done
export TMPDIR=/tmp
for cfg in B C; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${TAG}_prof_$cfg -o g -- python $ROOT/tools/gradient_bench.py --config $cfg --steps 4 --warmup 1 > /dev/null 2>&1)
  f=$(find gpurun_out/${TAG}_prof_$cfg -name "*kernel_stats.csv" | head -1); echo "== $cfg $f"; head -12 "$f" | cut -d, -f1-5 | cut -c1-150
done
