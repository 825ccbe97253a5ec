# round 6, LAB build: slice sizes re-swept with the ticket form (first-wave slice size x size above), and the launch traced
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export BEAGLE_MI355_ENGINE_LIB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
P=${1:-12500}
for top in 8 16 32; do for c in 40 56 70 96 128 150; do
  echo "patterns=$P chunk=$c top=$top $(BEAGLE_MI355_CHUNK=$c BEAGLE_MI355_CHUNK_TOP=$top timeout 150 python bench.py --steps 100 --patterns $P --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_us_per_eval'], d['ms_per_step_median'], d['roofline']['per_eval']['stored'])")"
done; done
BEAGLE_MI355_DUMP_PLAN=1 BEAGLE_MI355_WALK_TRACE=30 timeout 150 python bench.py --steps 40 --patterns $P --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>&1 | grep "mi355" | tail -45
