#!/bin/bash
# Round-5 A/B experiments, one GPU-box call each:   bash tools/r05_experiments.sh <what>      (results: profiles/r05_experiments.txt)
#   grad_ab    the 4-state gradient chain with every node stored / tip-tip nodes unstored / and those under one more tip
#   pad_ab     four-load fetches behind stores: never / behind write-mode rescaling / behind stored results too (A, A ALWAYS, shard)
#   gradtrace  rocprofv3 kernel trace of the gradient chain at 1e5 patterns
# The step and padding knobs exist in LAB builds only (beast-mcmc_amd/build.py --lab; csrc/kernels.h labEnv).
WHAT=${1:-grad_ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
LAB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
case $WHAT in
grad_ab)
  for P in 100000 20000; do
    for V in off 1 2; do
      if [ $V = off ]; then E="BEAGLE_MI355_GRADIENT_VIRTUAL=0"; else E="BEAGLE_MI355_GRADIENT_VIRTUAL=$V"; fi
      env $E BEAGLE_MI355_ENGINE_LIB=$LAB timeout 200 python tools/gradient_bench.py --patterns $P --steps 8 > gpurun_out/r5_grad_${P}_$V.json 2> gpurun_out/r5_grad_${P}_$V.err
      python - <<PY
import json
d=json.loads(open('gpurun_out/r5_grad_${P}_$V.json').read().strip().splitlines()[-1])
print('P=$P virtual=$V', d['ms_per_gradient'], 'ms; likelihood', d['ms_per_likelihood_same_driver'], 'ratio', d['gradient_over_likelihood'], 'stored', d['post_order_nodes_stored_per_gradient'])
PY
    done
  done;;
pad_ab)
  for V in 0 1 2; do
    export BEAGLE_MI355_WALK_PAD_FETCH=$V BEAGLE_MI355_ENGINE_LIB=$LAB
    timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > gpurun_out/r5_pad${V}_A.json 2>/dev/null
    timeout 300 python bench.py --rescaling always --steps 60 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > gpurun_out/r5_pad${V}_Aalways.json 2>/dev/null
    timeout 300 python bench.py --patterns 12500 --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > gpurun_out/r5_pad${V}_shard.json 2>/dev/null
    python - <<PY
import json
for n in ('A','Aalways','shard'):
    d=json.loads(open('gpurun_out/r5_pad${V}_%s.json' % n).read().strip().splitlines()[-1])
    print('pad=$V', n, d['value'], 'evals/s ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'kernel us', d['roofline']['kernel_us_per_eval'])
PY
  done;;
store_ab)
  # cache policy of the walk's result stores (tools/build_variant.sh <name> 'WALK4_STORE_POLICY=...' first): device-scope write-through
  # + non-temporal (the build's) / non-temporal only / plain / write-through only.  Without sc1 the slices of a one-launch program
  # may read stale lines across XCDs: TIMING ONLY.
  for V in product st_nt st_plain st_sc1; do
    if [ $V = product ]; then export BEAGLE_MI355_ENGINE_LIB=$R/beast-mcmc_amd/lib/libhmsbeagle-jni.so; else export BEAGLE_MI355_ENGINE_LIB=$R/build/variants/$V/libhmsbeagle-jni.so; fi
    [ -f $BEAGLE_MI355_ENGINE_LIB ] || continue
    timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > gpurun_out/r5_store_${V}_A.json 2>/dev/null
    timeout 300 python bench.py --patterns 12500 --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records > gpurun_out/r5_store_${V}_shard.json 2>/dev/null
    python - <<PY
import json
for n in ('A','shard'):
    d=json.loads(open('gpurun_out/r5_store_${V}_%s.json' % n).read().strip().splitlines()[-1])
    print('stores=$V', n, d['value'], 'evals/s median ms', d.get('ms_per_step_median'), 'kernel us', d['roofline']['kernel_us_per_eval'], 'lnL', d['lnL'])
PY
  done;;
gradtrace)
  (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5_gradtrace -o kt -- python $R/tools/gradient_bench.py --steps 6 > $R/gpurun_out/r5_gradtrace.json 2> $R/gpurun_out/r5_gradtrace.err)
  find gpurun_out/r5_gradtrace -name "*.db" -delete 2>/dev/null
  head -8 $(find gpurun_out/r5_gradtrace -name "*kernel_stats.csv" | head -1) | cut -c1-160;;
esac
