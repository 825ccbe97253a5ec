# (the knobs below exist in LAB builds only: python beast-mcmc_amd/build.py --lab)
export BEAGLE_MI355_ENGINE_LIB=${BEAGLE_MI355_ENGINE_LIB:-$(cd "$(dirname "$0")/.." && pwd)/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so}
# timing experiments on the walk kernel (results are WRONG by construction; only kernel_us_per_eval is of interest)
for a in 0 1 2 4 8 15; do
  echo "ablate=$a $(BEAGLE_MI355_ABLATE=$a timeout 150 python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o 'kernel_us_per_eval": [0-9.]*')"
done
