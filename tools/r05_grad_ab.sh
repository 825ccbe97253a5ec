#!/bin/bash
# Round 5, gradient chain A/B at 4 states: every node stored (round 4) / tip-tip nodes unstored / and those under one more tip (LAB build for the step knob)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
LAB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
for P in 100000 20000; do
  for V in off 1 2; do
    if [ $V = off ]; then E="BEAGLE_MI355_NO_GRADIENT_VIRTUAL=1"; else E="BEAGLE_MI355_GRADIENT_VIRTUAL_STEPS=$V"; fi
    env $E BEAGLE_MI355_ENGINE_LIB=$LAB timeout 200 python tools/gradient_bench.py --patterns $P --steps 8 > gpurun_out/r5_grad_${P}_$V.json 2> gpurun_out/r5_grad_${P}_$V.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/r5_grad_${P}_$V.json').read().strip().splitlines()[-1])
print('P=$P virtual=$V', d['ms_per_gradient'], 'ms; likelihood', d['ms_per_likelihood_same_driver'], 'ratio', d['gradient_over_likelihood'], 'stored', d['post_order_nodes_stored_per_gradient'])
PY
  done
done
for R_ in always; do
  timeout 300 python bench.py --rescaling always --steps 100 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > gpurun_out/r5_bench_A_always.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5_bench_A_always.json').read().strip().splitlines()[-1])
print('A ALWAYS', d['value'], 'evals/s kernel us', d['roofline']['kernel_us_per_eval'])
PY
done
