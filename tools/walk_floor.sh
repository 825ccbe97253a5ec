# (the knobs below exist in LAB builds only: python beast-mcmc_amd/build.py --lab)
export BEAGLE_MI355_ENGINE_LIB=${BEAGLE_MI355_ENGINE_LIB:-$(cd "$(dirname "$0")/.." && pwd)/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so}
#!/bin/bash
# What the walk's time is made of: rebuild the assembly loop with one part left out at a time (WALK4_EXPERIMENT, wrong
# results by construction) and time config A with and without the result stores.  Run on the GPU box (hipcc is there).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp beast-mcmc_amd/csrc/walk4_fast_loop.inc /tmp/walk4_fast_loop.inc.keep
for x in "" nofma notipread noinv nodma "notipread,noinv,nodma" "nofma,notipread,noinv,nodma"; do
  WALK4_EXPERIMENT=$x python tools/gen_walk4_fast.py > /dev/null; touch beast-mcmc_amd/csrc/kernels_walk4.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for a in 0 1; do
    echo "without [$x] stores $([ $a = 1 ] && echo off || echo on): $(BEAGLE_MI355_ABLATE=$a timeout 150 python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o 'kernel_us_per_eval": [0-9.]*')"
  done
done
cp /tmp/walk4_fast_loop.inc.keep beast-mcmc_amd/csrc/walk4_fast_loop.inc
