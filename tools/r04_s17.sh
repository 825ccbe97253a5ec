#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s17}
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
    print('%-28s %9.1f evals/s  %7.4f ms  kernel %7.1f us lnL %.9f' % ('${name}', d['value'], d['ms_per_step'], r['kernel_us_per_eval'], d['lnL']))
except Exception as e:
    print('${name} FAILED', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
V=BEAGLE_MI355_ENGINE_LIB=$ROOT/build/variants/nolateinv/libhmsbeagle-jni.so
for rep in 1 2; do
run shard_main_$rep A=1 -- --patterns 12500
run shard_nolate_$rep $V -- --patterns 12500
run p25k_main_$rep A=1 -- --patterns 25000
run p25k_nolate_$rep $V -- --patterns 25000
run A_main_$rep A=1 -- --steps 60
run A_nolate_$rep $V -- --steps 60
done
run A1wave_main BEAGLE_MI355_WALK_LDS_PAD=102400 -- --steps 40
run A1wave_nolate BEAGLE_MI355_WALK_LDS_PAD=102400 $V -- --steps 40
run A2wave_main BEAGLE_MI355_WALK_LDS_PAD=40960 -- --steps 40
run A2wave_nolate BEAGLE_MI355_WALK_LDS_PAD=40960 $V -- --steps 40
run D_main A=1 -- --config D
run D_nolate $V -- --config D
