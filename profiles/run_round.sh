#!/bin/bash
# Everything the round's measurement evidence comes from, in one GPU-box call:   bash profiles/run_round.sh r02
# bench lines (with the CPU baseline) for configs A-E under gpurun_out/<round>_bench_<config>.json, then the rocprofv3
# passes of profiles/collect.sh for A, B and C.  profiles/summarize.py (run afterwards, anywhere) files the summaries.
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for cfg in A B C D E; do
  steps=200; [ $cfg = B ] && steps=60; [ $cfg = C ] && steps=60
  timeout 300 python bench.py --config $cfg --steps $steps --warmup 5 > gpurun_out/${R}_bench_$cfg.json 2> gpurun_out/${R}_bench_$cfg.err
  echo "bench $cfg rc=$? $(python -c "import json;d=json.loads(open('gpurun_out/${R}_bench_$cfg.json').read().strip().splitlines()[-1]);print(d['value'],'evals/s frac',d['roofline']['frac'],'cpu',d['cpu_baseline'] and d['cpu_baseline']['value'])" 2>&1 | tail -1)"
done
timeout 120 python bench.py --config A --caller btl --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/${R}_bench_A_btl.json 2>/dev/null
TAG=_A STEPS=10 bash profiles/collect.sh $R --config A 2>&1 | tail -5
TAG=_B STEPS=5 bash profiles/collect.sh $R --config B 2>&1 | tail -5
TAG=_C STEPS=5 bash profiles/collect.sh $R --config C 2>&1 | tail -5
