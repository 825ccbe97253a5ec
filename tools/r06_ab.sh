# round 6: A/B of one product switch on the 4-state configurations, alternating passes:  bash tools/r06_ab.sh BEAGLE_MI355_NO_LOAD_SKIP [passes]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
VAR=$1; PASSES=${2:-2}
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; pe=r.get('per_eval') or {}; print(d['value'], 'evals/s  ms/step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), ' kernel us', r['kernel_us_per_eval'], ' fused', pe.get('fused_cherries'), ' lnL', repr(d['lnL']))"; }
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
for pass in $(seq 1 $PASSES); do for t in 0 1; do
  echo "== pass $pass $VAR=$t"
  export $VAR=$t
  echo "A: $(timeout 300 python bench.py --steps 100 --warmup 10 $common 2>/dev/null | line)"
  echo "A always: $(timeout 300 python bench.py --rescaling always --steps 60 --warmup 5 $common 2>/dev/null | line)"
  echo "shard 12500 (sharded path): $(timeout 200 python bench.py --patterns 12500 --force-sharded --steps 200 --warmup 12 $common 2>/dev/null | line)"
  echo "D real1: $(timeout 300 python bench.py --real benchmark1 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "D real2: $(timeout 300 python bench.py --real benchmark2 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "E: $(timeout 300 python bench.py --config E --steps 300 --warmup 20 --no-cpu-baseline --no-live-traffic --no-side-records 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'], repr(d['lnL']))")"
done; done
