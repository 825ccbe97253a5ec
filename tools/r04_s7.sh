#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s7}
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_${name}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-28s %9.1f evals/s  %7.4f ms  kernel %7.1f us  other %s' % ('${name}', d['value'], d['ms_per_step'], r['kernel_us_per_eval'], d.get('other_caller')))
except Exception as e:
    print('${name} FAILED', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
run btl_default A=1 -- --caller btl --steps 50
run btl_copyengine BEAGLE_MI355_COPY_ENGINE_UPLOADS=1 -- --caller btl --steps 50
run btl_nofusion BEAGLE_MI355_NO_WALK_FUSION=1 -- --caller btl --steps 50
run A_w5_s20 A=1 -- --steps 20 --warmup 5
run A_w100_s20 A=1 -- --steps 20 --warmup 100
run A_w5_s200 A=1 -- --steps 200 --warmup 5
run A_w5_s20_again A=1 -- --steps 20 --warmup 5
for pad in 0 16384 40960 102400; do run A_ldspad$pad BEAGLE_MI355_WALK_LDS_PAD=$pad -- --steps 60; done
BEAGLE_MI355_DUMP_PLAN=2 timeout 200 python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2> gpurun_out/${TAG}_A_plan_dump.txt >/dev/null; grep -c "k1" gpurun_out/${TAG}_A_plan_dump.txt
python - <<'PY'
# keep ONE read-mode program (the last complete dump)
import re
txt=open('gpurun_out/r04s7_A_plan_dump.txt').read().split('[mi355] plan:')
last=txt[-1]
open('gpurun_out/r04s7_A_plan_last.txt','w').write('[mi355] plan:'+last)
print(len(re.findall(r'k1 \d', last)), 'micro-ops in the last dumped program')
PY
timeout 200 python bench.py --patterns 12500 --no-cpu-baseline --no-live-traffic --no-library-route > gpurun_out/${TAG}_shard_line.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_shard_line.json').read().strip().splitlines()[-1]); print('shard', d['value'], d['ms_per_step'], d.get('partial_update'))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/${TAG}_A_line.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_A_line.json').read().strip().splitlines()[-1]); print('A', d['value'], d['ms_per_step'], d.get('shard_point'))
PY
