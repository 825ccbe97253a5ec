# round 6, LAB build: slice sizes for the alignments of a few pattern groups (the reference's benchmark1 / benchmark2, config E), on tickets
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export BEAGLE_MI355_ENGINE_LIB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
k() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_us_per_eval'], d.get('ms_per_step_median') or d['ms_per_step'])"; }
for top in 6 8 12 16; do for c in 0 12 16 24 32 48; do
  export BEAGLE_MI355_CHUNK_TOP=$top
  if [ $c = 0 ]; then unset BEAGLE_MI355_CHUNK; else export BEAGLE_MI355_CHUNK=$c; fi
  echo "chunk=$c top=$top   D1: $(timeout 200 python bench.py --real benchmark1 --steps 200 --warmup 20 $common 2>/dev/null | k)   D2: $(timeout 200 python bench.py --real benchmark2 --steps 200 --warmup 20 $common 2>/dev/null | k)   E: $(timeout 200 python bench.py --config E --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-side-records 2>/dev/null | k)"
done; done
