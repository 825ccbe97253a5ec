#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s26}
timeout 600 python -m pytest tests/test_gpu_root_fusion.py -m gpu -x -q > gpurun_out/${TAG}_pytest1.log 2>&1; echo "fusion tests rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest1.log | tail -1)"; grep -E "^FAILED|^E  " gpurun_out/${TAG}_pytest1.log | head -12
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -1)"; grep -E "^FAILED|^E  " gpurun_out/${TAG}_pytest.log | head -12
run() { # label, env...
  local label=$1; shift
  echo "$label: $(env "$@" timeout 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'evals/s', d['ms_per_step'], 'ms kernel', r['kernel_us_per_eval'], 'lnL', d['lnL'])")"
}
EXTRA="--config A --patterns 12500"; run "shard fused" X=1; run "shard unfused" BEAGLE_MI355_NO_ROOT_FUSION=1; run "shard fused" X=1; run "shard unfused" BEAGLE_MI355_NO_ROOT_FUSION=1
EXTRA="--config A --patterns 12500 --force-sharded"; run "shard sharded-path fused" X=1; run "shard sharded-path unfused" BEAGLE_MI355_NO_ROOT_FUSION=1
EXTRA="--config D"; run "D fused" X=1; run "D unfused" BEAGLE_MI355_NO_ROOT_FUSION=1
EXTRA="--config A --steps 100"; run "A fused" X=1; run "A unfused" BEAGLE_MI355_NO_ROOT_FUSION=1
