#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s25}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -1)"; grep -E "^FAILED|^E  " gpurun_out/${TAG}_pytest.log | head -12
D1=$ROOT/build/variants/depth1/libhmsbeagle-jni.so
for cfg in B C; do
  for lib in "" $D1; do echo "$cfg lib=[$lib]: $(BEAGLE_MI355_ENGINE_LIB=$lib timeout 300 python bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'], d['lnL'])")"; done
done
