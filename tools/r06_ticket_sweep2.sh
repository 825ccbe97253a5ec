# round 6, LAB build: the first-wave rule's divisor x the slice size above, on tickets, for the small workloads
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export BEAGLE_MI355_ENGINE_LIB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
k() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_us_per_eval'], d.get('ms_per_step_median') or d['ms_per_step'])"; }
for top in 8 16; do for div in 500 765 1000 1400 2000; do
  export BEAGLE_MI355_CHUNK_DIV=$div BEAGLE_MI355_CHUNK_TOP=$top
  echo "div=$div top=$top  E: $(timeout 200 python bench.py --config E --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-side-records 2>/dev/null | k)   D1: $(timeout 200 python bench.py --real benchmark1 --steps 200 --warmup 20 $common 2>/dev/null | k)   D2: $(timeout 200 python bench.py --real benchmark2 --steps 200 --warmup 20 $common 2>/dev/null | k)   6250: $(timeout 200 python bench.py --patterns 6250 --steps 200 --warmup 20 $common 2>/dev/null | k)   12500: $(timeout 200 python bench.py --patterns 12500 --steps 200 --warmup 20 $common 2>/dev/null | k)   25000: $(timeout 200 python bench.py --patterns 25000 --steps 100 --warmup 20 $common 2>/dev/null | k)"
done; done
