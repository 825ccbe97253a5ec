#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s4}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/${TAG}_pytest.log)"
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_${name}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-28s %9.1f evals/s  %7.4f ms  kernel %7.1f us  stored %s lnL %.6f' % ('${name}', d['value'], d['ms_per_step'], r['kernel_us_per_eval'], r.get('per_eval',{}).get('stored'), d['lnL']))
except Exception as e:
    print('${name} FAILED', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
for c in 40 56 72 96 128; do for t in 0 8 16; do run shard_c${c}_t$t BEAGLE_MI355_CHUNK=$c BEAGLE_MI355_CHUNK_TOP=$t -- --patterns 12500; done; done
run shard_c56_t8_v16 BEAGLE_MI355_CHUNK=56 BEAGLE_MI355_CHUNK_TOP=8 BEAGLE_MI355_VSTEPS=16 -- --patterns 12500
run shard_c56_t8_v24 BEAGLE_MI355_CHUNK=56 BEAGLE_MI355_CHUNK_TOP=8 BEAGLE_MI355_VSTEPS=24 -- --patterns 12500
for t in 0 8 16 32; do run A_t$t BEAGLE_MI355_CHUNK_TOP=$t -- --steps 100; done
run A_c200_t16 BEAGLE_MI355_CHUNK=200 BEAGLE_MI355_CHUNK_TOP=16 -- --steps 100
run A_c300_t16 BEAGLE_MI355_CHUNK=300 BEAGLE_MI355_CHUNK_TOP=16 -- --steps 100
for t in 0 8 16; do run p25k_t$t BEAGLE_MI355_CHUNK_TOP=$t -- --patterns 25000; done
run p25k_c100_t8 BEAGLE_MI355_CHUNK=100 BEAGLE_MI355_CHUNK_TOP=8 -- --patterns 25000
run p25k_c150_t8 BEAGLE_MI355_CHUNK=150 BEAGLE_MI355_CHUNK_TOP=8 -- --patterns 25000
for t in 0 8; do run p50k_t$t BEAGLE_MI355_CHUNK_TOP=$t -- --patterns 50000; done
for t in 0 8; do run E_t$t BEAGLE_MI355_CHUNK_TOP=$t -- --config E; done
for t in 0 8; do run D_t$t BEAGLE_MI355_CHUNK_TOP=$t -- --config D; done
(cd /tmp && export TMPDIR=/tmp && BEAGLE_MI355_CHUNK=56 BEAGLE_MI355_CHUNK_TOP=8 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_trace" -o kt -- \
   python "$ROOT/bench.py" --patterns 12500 --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > "$ROOT/gpurun_out/${TAG}_trace.json" 2> "$ROOT/gpurun_out/${TAG}_trace.err"; echo "trace rc=$?")
find gpurun_out/${TAG}_trace -name "*.db" -delete 2>/dev/null
python tools/timeline.py gpurun_out/${TAG}_trace 2>&1 | tail -14
