#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s22}
timeout 1200 python -m pytest tests/test_gpu_walk_kernels.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -1)"; grep -E "^FAILED|^E  " gpurun_out/${TAG}_pytest.log | head -12
run() { # label, env..., -- bench args
  local label=$1; shift
  echo "$label: $(env "$@" timeout 200 python bench.py --config A --steps 60 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'evals/s', d['ms_per_step'], 'ms kernel', r['kernel_us_per_eval'], 'lnL', d['lnL'])")"
}
D1=$ROOT/build/variants/depth1/libhmsbeagle-jni.so
EXTRA=""; run "A depth2" X=1; run "A depth1" BEAGLE_MI355_ENGINE_LIB=$D1; run "A depth2 lax" BEAGLE_MI355_STRICT_WAITS=0; run "A depth1 lax" BEAGLE_MI355_ENGINE_LIB=$D1 BEAGLE_MI355_STRICT_WAITS=0; run "A depth2" X=1
EXTRA="--patterns 12500 --steps 200"; run "shard depth2" X=1; run "shard depth1" BEAGLE_MI355_ENGINE_LIB=$D1; run "shard depth2 lax" BEAGLE_MI355_STRICT_WAITS=0
EXTRA="--rescaling always"; run "A always depth2" X=1; run "A always depth1" BEAGLE_MI355_ENGINE_LIB=$D1
