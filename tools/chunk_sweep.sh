# (the knobs below exist in LAB builds only: python beast-mcmc_amd/build.py --lab)
export BEAGLE_MI355_ENGINE_LIB=${BEAGLE_MI355_ENGINE_LIB:-$(cd "$(dirname "$0")/.." && pwd)/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so}
# how the size of the independent subtrees the planner cuts the forest into (BEAGLE_MI355_CHUNK, 0 = one walk) moves the headline
# usage: bash tools/chunk_sweep.sh [patterns] [chunk sizes...]
P=${1:-100000}; shift
for c in ${@:-0 600 300 150 80}; do
  echo "patterns=$P chunk=$c $(BEAGLE_MI355_CHUNK=$c timeout 150 python bench.py --steps 30 --patterns $P --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*\|kernel_us_per_eval": [0-9.]*\|"stored": [0-9.]*\|"mem_reads": [0-9.]*\|"walks": [0-9.]*' | tr '\n' ' ')"
done
