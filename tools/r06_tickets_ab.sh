# round 6: tickets (retuned slice sizes) against flags, alternating, two passes: bash tools/r06_tickets_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'evals/s  ms/step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), ' kernel us', r['kernel_us_per_eval'], ' stored', (r.get('per_eval') or {}).get('stored'))"; }
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
for pass in 1 2; do for t in 0 1; do
  echo "== pass $pass NO_WALK_TICKETS=$t"
  for p in 6250 12500 25000; do
    echo "shard $p (sharded path): $(BEAGLE_MI355_NO_WALK_TICKETS=$t timeout 200 python bench.py --patterns $p --force-sharded --steps 200 --warmup 12 $common 2>/dev/null | line)"
  done
  echo "A: $(BEAGLE_MI355_NO_WALK_TICKETS=$t timeout 300 python bench.py --steps 60 --warmup 10 $common 2>/dev/null | line)"
  echo "D real1: $(BEAGLE_MI355_NO_WALK_TICKETS=$t timeout 300 python bench.py --real benchmark1 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "D real2: $(BEAGLE_MI355_NO_WALK_TICKETS=$t timeout 300 python bench.py --real benchmark2 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "E: $(BEAGLE_MI355_NO_WALK_TICKETS=$t timeout 300 python bench.py --config E --steps 300 --warmup 20 --no-cpu-baseline --no-live-traffic --no-side-records 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'])")"
done; done
