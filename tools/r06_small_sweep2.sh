# round 6, LAB build: slice sizes for benchmark1 / E again (after the cap and cut-off changes), two passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export BEAGLE_MI355_ENGINE_LIB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
k() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_us_per_eval'], d.get('ms_per_step_median') or d['ms_per_step'])"; }
for pass in 1 2; do for top in 8 12 16 24; do for c in 0 16 32 48 64; do
  export BEAGLE_MI355_CHUNK_TOP=$top
  if [ $c = 0 ]; then unset BEAGLE_MI355_CHUNK; else export BEAGLE_MI355_CHUNK=$c; fi
  echo "pass $pass chunk=$c top=$top   D1: $(timeout 200 python bench.py --real benchmark1 --steps 300 --warmup 20 $common 2>/dev/null | k)   D2: $(timeout 200 python bench.py --real benchmark2 --steps 300 --warmup 20 $common 2>/dev/null | k)"
done; done; done
