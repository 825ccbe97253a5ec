#!/bin/bash
# Round 5: fetch padding behind stores (engine_walk.cpp runPlan), LAB build: 0 / 1 / 2; config A read mode (driver command line), ALWAYS, the 12 500-pattern shard
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
LAB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
for V in 0 1 2; do
  export BEAGLE_MI355_WALK_PAD_FETCH=$V BEAGLE_MI355_ENGINE_LIB=$LAB
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > gpurun_out/r5_pad${V}_A.json 2>/dev/null
  timeout 300 python bench.py --rescaling always --steps 60 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > gpurun_out/r5_pad${V}_Aalways.json 2>/dev/null
  timeout 300 python bench.py --patterns 12500 --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > gpurun_out/r5_pad${V}_shard.json 2>/dev/null
  python - <<PY
import json
for n in ('A','Aalways','shard'):
    d=json.loads(open('gpurun_out/r5_pad${V}_%s.json' % n).read().strip().splitlines()[-1])
    print('pad=$V', n, d['value'], 'evals/s ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'kernel us', d['roofline']['kernel_us_per_eval'])
PY
done
