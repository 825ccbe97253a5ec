# round 6: the XCD-aware row layout for launches the chip holds at once (BEAGLE_MI355_NO_XCD_MAP=1 = plain grid), E and the small 4-state sizes, with E's in-run traffic
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'evals/s  median ms', d.get('ms_per_step_median'), ' kernel us', r['kernel_us_per_eval'], ' traffic', r.get('traffic'), ' lnL', repr(d['lnL']))"; }
for pass in 1 2; do for t in 0 1; do
  export BEAGLE_MI355_NO_XCD_MAP=$t
  echo "== pass $pass NO_XCD_MAP=$t"
  echo "E (with traffic passes): $(timeout 300 python bench.py --config E --steps 300 --warmup 20 --no-cpu-baseline --no-side-records 2>/dev/null | line)"
  echo "D real1: $(timeout 300 python bench.py --real benchmark1 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "D real2: $(timeout 300 python bench.py --real benchmark2 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "shard 6250: $(timeout 200 python bench.py --patterns 6250 --steps 300 --warmup 12 $common 2>/dev/null | line)"
  echo "A: $(timeout 300 python bench.py --steps 40 --warmup 10 $common 2>/dev/null | line)"
done; done
