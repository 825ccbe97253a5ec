# (the knobs below exist in LAB builds only: python beast-mcmc_amd/build.py --lab)
export BEAGLE_MI355_ENGINE_LIB=${BEAGLE_MI355_ENGINE_LIB:-$(cd "$(dirname "$0")/.." && pwd)/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so}
#!/bin/bash
# k_walk4 (the C++ walk kernel: write-mode rescaling, unaligned partitions) with and without the non-temporal hint on its
# half-line result stores; run on the GPU box.  BEAGLE_MI355_NO_FAST_WALK=1 sends every launch to it.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp beast-mcmc_amd/csrc/kernels_walk4.hip /tmp/kernels_walk4.hip.keep
for pol in nt default; do
  cp /tmp/kernels_walk4.hip.keep beast-mcmc_amd/csrc/kernels_walk4.hip
  [ $pol = default ] && sed -i 's/%\[base\] nt\\n\\t"/%[base]\\n\\t"/; s/%\[base\] offset:16 nt\\n\\t"/%[base] offset:16\\n\\t"/' beast-mcmc_amd/csrc/kernels_walk4.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "k_walk4 stores $pol: $(BEAGLE_MI355_NO_FAST_WALK=1 timeout 150 python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*\|kernel_us_per_eval": [0-9.]*' | tr '\n' ' ') | config E: $(timeout 150 python bench.py --config E --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*\|kernel_us_per_eval": [0-9.]*' | tr '\n' ' ')"
done
cp /tmp/kernels_walk4.hip.keep beast-mcmc_amd/csrc/kernels_walk4.hip
