#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s24}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -1)"; grep -E "^FAILED|^E  " gpurun_out/${TAG}_pytest.log | head -12
run() { # label, env...
  local label=$1; shift
  echo "$label: $(env "$@" timeout 200 python bench.py --config A --steps 60 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'evals/s', d['ms_per_step'], 'ms kernel', r['kernel_us_per_eval'], 'lnL', d['lnL'])")"
}
D1=$ROOT/build/variants/depth1/libhmsbeagle-jni.so
EXTRA=""; run "A depth2" X=1; run "A depth1" BEAGLE_MI355_ENGINE_LIB=$D1; run "A depth2" X=1; run "A depth1" BEAGLE_MI355_ENGINE_LIB=$D1
EXTRA="--patterns 12500 --steps 200"; run "shard depth2" X=1; run "shard depth1" BEAGLE_MI355_ENGINE_LIB=$D1; run "shard depth2" X=1; run "shard depth1" BEAGLE_MI355_ENGINE_LIB=$D1
EXTRA="--patterns 25000 --steps 200"; run "25k depth2" X=1; run "25k depth1" BEAGLE_MI355_ENGINE_LIB=$D1
EXTRA="--config D --steps 200"; run "D depth2" X=1; run "D depth1" BEAGLE_MI355_ENGINE_LIB=$D1
EXTRA="--config E --steps 200"
for lib in "" $D1; do echo "E lib=[$lib]: $(BEAGLE_MI355_ENGINE_LIB=$lib timeout 200 python bench.py --config E --steps 200 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'])")"; done
