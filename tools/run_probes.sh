#!/bin/bash
# Build and run every hardware probe of tools/ on the GPU box; output -> gpurun_out/<round>_probes.txt (copy to profiles/)
R=${1:-r02}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${R}_probes.txt; mkdir -p $ROOT/gpurun_out; : > $OUT
for p in mfma_f64_probe walk_probe vmem_rate_probe glds_probe hbm_write_probe; do
  echo "==== tools/$p.hip" >> $OUT
  if hipcc --offload-arch=gfx950 -O3 -std=c++17 $ROOT/tools/$p.hip -o /tmp/$p 2>/dev/null; then timeout 120 /tmp/$p >> $OUT 2>&1 || echo "(exit $?)" >> $OUT; else echo "(build failed)" >> $OUT; fi
done
cat $OUT
