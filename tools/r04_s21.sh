#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s21}
for a in 0 1 4 8 12 13 15 0; do
  echo "A ablate=$a: $(BEAGLE_MI355_ABLATE=$a timeout 200 python bench.py --config A --steps 60 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'evals/s kernel', r['kernel_us_per_eval'])")"
done
for a in 0 1 13; do
  echo "shard ablate=$a: $(BEAGLE_MI355_ABLATE=$a timeout 200 python bench.py --config A --patterns 12500 --steps 100 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'evals/s kernel', r['kernel_us_per_eval'])")"
done
for c in 32 64 128 256 512; do
  echo "A gradient PRE_CHUNK=$c: $(BEAGLE_MI355_PRE_CHUNK=$c timeout 300 python tools/gradient_bench.py --config A --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_gradient'], 'ms; likelihood', d['ms_per_likelihood_same_driver'], d['how'])")"
done
