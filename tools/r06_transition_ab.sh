# round 6: two engine libraries against each other where the transition kernel weighs (small alignments, the shard), alternating, with the
# transition kernel's own average from a kernel trace:   bash tools/r06_transition_ab.sh <a.so> <b.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'evals/s  median ms', d.get('ms_per_step_median'), ' kernel us', r['kernel_us_per_eval'], ' lnL', repr(d['lnL']))"; }
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
for pass in 1 2; do for L in "$@"; do
  export BEAGLE_MI355_ENGINE_LIB=$R/$L
  echo "== pass $pass $L"
  echo "D real1: $(timeout 300 python bench.py --real benchmark1 --steps 400 --warmup 20 $common 2>/dev/null | line)"
  echo "D real2: $(timeout 300 python bench.py --real benchmark2 --steps 400 --warmup 20 $common 2>/dev/null | line)"
  echo "shard:   $(timeout 300 python bench.py --patterns 12500 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "A:       $(timeout 300 python bench.py --steps 100 --warmup 10 $common 2>/dev/null | line)"
done; done
for L in "$@"; do
  export BEAGLE_MI355_ENGINE_LIB=$R/$L
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tk && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk -o kt -- python $R/bench.py --real benchmark1 --steps 200 --warmup 20 $common > /dev/null 2>&1)
  echo "== $L kernel averages (ns), benchmark1"; python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/tk/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('transition4Fused','gatherAndSnapshot','walk4_fast')): print('  ', r['Name'][:40], r['Calls'], r['AverageNs'])
PY
done
