# first-wave slice size x the size of the slices above it, at one shard size (LAB build): bash tools/chunk_sweep2.sh [patterns]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export BEAGLE_MI355_ENGINE_LIB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
P=${1:-12500}
for top in 16 32 64; do for c in 40 56 70 96 128 150; do
  echo "patterns=$P chunk=$c top=$top $(BEAGLE_MI355_CHUNK=$c BEAGLE_MI355_CHUNK_TOP=$top timeout 150 python bench.py --steps 100 --patterns $P --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_us_per_eval'], d['ms_per_step_median'], d['roofline']['per_eval']['stored'])")"
done; done
