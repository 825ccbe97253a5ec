# round 6: two engine libraries against each other on config B (20 states), DYNAMIC steady state and scheme ALWAYS, alternating: bash tools/r06_lib_ab_B.sh <a.so> <b.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'evals/s  ms/step', d['ms_per_step'], ' kernel us', r['kernel_us_per_eval'], ' lnL', repr(d['lnL']))"; }
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records --no-other-configs"
for pass in 1 2; do for L in "$@"; do
  export BEAGLE_MI355_ENGINE_LIB=$R/$L
  echo "== pass $pass $L"
  echo "B dynamic: $(timeout 300 python bench.py --config B --steps 40 --warmup 5 $common 2>/dev/null | line)"
  echo "B always:  $(timeout 300 python bench.py --config B --rescaling always --steps 30 --warmup 5 $common 2>/dev/null | line)"
done; done
