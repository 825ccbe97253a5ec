#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s20}
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config_e" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -1)"; grep -E "^FAILED|Error|assert" gpurun_out/${TAG}_pytest.log | head -8
for r in none dynamic; do
  timeout 300 python bench.py --config A --rescaling $r --steps 100 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('A rescaling=$r', d['value'], d['ms_per_step'], 'kernel', r['kernel_us_per_eval'], r['per_eval'], d['lnL'])"
done
timeout 300 python bench.py --config E --no-cpu-baseline --no-live-traffic --no-library-route 2>gpurun_out/${TAG}_E.err | tail -1 > gpurun_out/${TAG}_bench_E.json; python -c "import json; d=json.loads(open('gpurun_out/${TAG}_bench_E.json').read()); print('E', d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'], d['partial_update'])"; tail -3 gpurun_out/${TAG}_E.err
for cfg in B C; do
  timeout 600 python tools/gradient_bench.py --config $cfg --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${TAG}_gradient_bench_$cfg.json
  python -c "import json; d=json.loads(open('gpurun_out/${TAG}_gradient_bench_$cfg.json').read()); print('$cfg', d['ms_per_gradient'], 'ms; likelihood', d['ms_per_likelihood_same_driver'], d['roofline'], d['how'])"
done
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof_A -o g -- python $ROOT/tools/gradient_bench.py --config A --steps 4 --warmup 1 > /dev/null 2>&1)
f=$(find gpurun_out/${TAG}_prof_A -name "*kernel_stats.csv" | head -1); echo "== A gradient $f"; head -10 "$f" | cut -d, -f1-4 | cut -c1-160
find gpurun_out/${TAG}_prof_A -name "*.db" -delete 2>/dev/null
