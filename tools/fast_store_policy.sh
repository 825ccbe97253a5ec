# (the knobs below exist in LAB builds only: python beast-mcmc_amd/build.py --lab)
export BEAGLE_MI355_ENGINE_LIB=${BEAGLE_MI355_ENGINE_LIB:-$(cd "$(dirname "$0")/.." && pwd)/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so}
#!/bin/bash
# the assembly loop's full-line result stores with and without the non-temporal hint; run on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp beast-mcmc_amd/csrc/walk4_fast_loop.inc /tmp/walk4_fast_loop.inc.keep
for pol in " nt" ""; do
  WALK4_STORE_POLICY="$pol" python tools/gen_walk4_fast.py > /dev/null
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  r() { timeout 150 python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*\|kernel_us_per_eval": [0-9.]*' | tr '\n' ' '; }
  echo "k_walk4_fast stores [$pol]: A $(r --steps 60) | 12 500 patterns $(r --steps 100 --patterns 12500) | D $(r --config D --steps 100)"
done
cp /tmp/walk4_fast_loop.inc.keep beast-mcmc_amd/csrc/walk4_fast_loop.inc
