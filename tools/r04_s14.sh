#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r04s14}
for cfg in B C; do
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_grad$cfg" -o kt -- \
   python "$ROOT/tools/gradient_bench.py" --config $cfg --steps 3 > "$ROOT/gpurun_out/${TAG}_grad$cfg.json" 2> "$ROOT/gpurun_out/${TAG}_grad$cfg.err"; echo "rc=$?")
find gpurun_out/${TAG}_grad$cfg -name "*.db" -delete 2>/dev/null
python - <<PY
import csv,glob
for f in glob.glob('gpurun_out/${TAG}_grad$cfg/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:12]: print('$cfg', r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
done
BEAGLE_MI355_DUMP_PLAN=1 timeout 200 python bench.py --patterns 12500 --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>&1 >/dev/null | grep "plan:" | sort | uniq -c | head -3
