# round 6: two engine libraries against each other on the latency-bound sizes, alternating: bash tools/r06_lib_ab.sh build/variants/base/libhmsbeagle-jni.so build/variants/warm/libhmsbeagle-jni.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'evals/s  median ms', d.get('ms_per_step_median'), ' kernel us', r['kernel_us_per_eval'], ' lnL', repr(d['lnL']))"; }
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
for pass in 1 2 3; do for L in "$@"; do
  export BEAGLE_MI355_ENGINE_LIB=$R/$L
  echo "== pass $pass $L"
  echo "shard 12500 (sharded path): $(timeout 200 python bench.py --patterns 12500 --force-sharded --steps 300 --warmup 12 $common 2>/dev/null | line)"
  echo "shard 6250: $(timeout 200 python bench.py --patterns 6250 --force-sharded --steps 300 --warmup 12 $common 2>/dev/null | line)"
  echo "D real1: $(timeout 300 python bench.py --real benchmark1 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "A: $(timeout 300 python bench.py --steps 60 --warmup 10 $common 2>/dev/null | line)"
done; done
