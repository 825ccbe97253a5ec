# round 6: the one-launch walk on tickets against the flag form (same build, BEAGLE_MI355_NO_WALK_TICKETS=1): bash tools/r06_tickets.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'evals/s  ms/step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), ' kernel us', r['kernel_us_per_eval'], ' stored', (r.get('per_eval') or {}).get('stored'), ' lnL', repr(d['lnL']))"; }
for t in 0 1; do
  echo "== NO_WALK_TICKETS=$t"
  for p in 12500 25000 50000; do
    echo "shard $p (sharded path): $(BEAGLE_MI355_NO_WALK_TICKETS=$t timeout 200 python bench.py --patterns $p --force-sharded --steps 200 --warmup 12 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | line)"
  done
  echo "A: $(BEAGLE_MI355_NO_WALK_TICKETS=$t timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | line)"
  echo "A always: $(BEAGLE_MI355_NO_WALK_TICKETS=$t timeout 300 python bench.py --rescaling always --steps 60 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | line)"
  echo "D real1: $(BEAGLE_MI355_NO_WALK_TICKETS=$t timeout 300 python bench.py --real benchmark1 --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | line)"
  echo "E: $(BEAGLE_MI355_NO_WALK_TICKETS=$t timeout 300 python bench.py --config E --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-side-records 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'], repr(d['lnL']))")"
done
