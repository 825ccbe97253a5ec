# round 6: two engine libraries against each other where the matrix gather weighs (shard, A, the small alignments, E), alternating,
# with per-kernel statistics of the last pass: bash tools/r06_gather_ab.sh <a.so> <b.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'evals/s  ms/step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), ' kernel us', r['kernel_us_per_eval'], ' lnL', repr(d['lnL']))"; }
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
for pass in 1 2 3; do for L in "$@"; do
  export BEAGLE_MI355_ENGINE_LIB=$R/$L
  echo "== pass $pass $L"
  echo "shard 12500: $(timeout 200 python bench.py --patterns 12500 --force-sharded --steps 300 --warmup 12 $common 2>/dev/null | line)"
  echo "A: $(timeout 300 python bench.py --steps 80 --warmup 10 $common 2>/dev/null | line)"
  echo "D real1: $(timeout 300 python bench.py --real benchmark1 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "D real2: $(timeout 300 python bench.py --real benchmark2 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "E: $(timeout 300 python bench.py --config E --steps 300 --warmup 20 --no-cpu-baseline --no-live-traffic --no-side-records 2>/dev/null | line)"
done; done
cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  export BEAGLE_MI355_ENGINE_LIB=$R/$L
  n=$(basename $L .so)
  for c in E A; do
    extra=""; [ $c = E ] && extra="--config E"
    rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gather_ab/$n$c -o p -- python $R/bench.py $extra --steps 60 --warmup 10 --no-cpu-baseline --no-live-traffic --no-side-records --no-other-configs > /dev/null 2>&1
    echo "== $n $c kernel stats"; python - <<PY
import sqlite3
db = sqlite3.connect("$R/gpurun_out/gather_ab/$n$c/p_results.db")
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
for r in db.execute("select s.kernel_name, count(*), avg(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by sum(d.end-d.start) desc limit 7" % (kd, ks)):
    print("   %-56s %5d calls  %8.2f us" % (r[0][:56], r[1], r[2] / 1000))
PY
  done
done
