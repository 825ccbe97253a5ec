#!/bin/bash
# round-4 session 1: where the 12 500-pattern shard's step goes (current code), and the library route's step-time spread
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s1}
BTL_TIMING=1 BEAGLE_MI355_HOST_TIMING=1 timeout 200 python tools/step_profile.py 12500 > gpurun_out/${TAG}_host.txt 2>&1; tail -25 gpurun_out/${TAG}_host.txt
BEAGLE_MI355_DUMP_PLAN=1 timeout 200 python bench.py --patterns 12500 --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-library-route 2>&1 >/dev/null | grep "plan:" | sort | uniq -c | head -8
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_trace" -o kt -- \
   python "$ROOT/bench.py" --patterns 12500 --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --no-library-route > "$ROOT/gpurun_out/${TAG}_trace.json" 2> "$ROOT/gpurun_out/${TAG}_trace.err"; echo "trace rc=$?")
find gpurun_out/${TAG}_trace -name "*.db" -delete 2>/dev/null
python tools/timeline.py gpurun_out/${TAG}_trace 2>&1 | tail -24
timeout 300 python bench.py --patterns 12500 --no-cpu-baseline --no-live-traffic > gpurun_out/${TAG}_bench_shard.json 2> gpurun_out/${TAG}_bench_shard.err; echo "shard rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_shard.json').read().strip().splitlines()[-1])
print('shard12500', d['value'], 'evals/s ms', d['ms_per_step'], 'kernel us', d['roofline']['kernel_us_per_eval'], 'lib', d.get('library_route'))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/${TAG}_bench_A20.json 2> gpurun_out/${TAG}_bench_A20.err; echo "A20 rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_A20.json').read().strip().splitlines()[-1])
print('A 20 steps', d['value'], 'evals/s ms', d['ms_per_step'], 'lib', d.get('library_route'))
PY
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
