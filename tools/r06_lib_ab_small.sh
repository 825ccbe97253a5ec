# round 6: two engine libraries against each other on the small alignments, alternating: bash tools/r06_lib_ab_small.sh <a.so> <b.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'evals/s  median ms', d.get('ms_per_step_median'), ' kernel us', r['kernel_us_per_eval'], ' lnL', repr(d['lnL']))"; }
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
for pass in 1 2 3; do for L in "$@"; do
  export BEAGLE_MI355_ENGINE_LIB=$R/$L
  echo "== pass $pass $L"
  echo "D real1: $(timeout 300 python bench.py --real benchmark1 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "D real2: $(timeout 300 python bench.py --real benchmark2 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "D synth: $(timeout 300 python bench.py --config D --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "E: $(timeout 300 python bench.py --config E --steps 300 --warmup 20 --no-cpu-baseline --no-live-traffic --no-side-records 2>/dev/null | line)"
done; done
