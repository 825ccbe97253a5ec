R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
LAB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
for CH in default 128 64 32 16; do
  if [ $CH = default ]; then E="X=1"; else E="BEAGLE_MI355_PRE_CHUNK=$CH"; fi
  env $E BEAGLE_MI355_HOST_TIMING=2 BEAGLE_MI355_ENGINE_LIB=$LAB timeout 200 python tools/gradient_bench.py --patterns 100000 --steps 8 > /tmp/g.json 2> /tmp/g.err
  grep "pre-order walk" /tmp/g.err | tail -1
  python - <<PY
import json
d=json.loads(open('/tmp/g.json').read().strip().splitlines()[-1])
print('chunk=$CH', d['ms_per_gradient'], 'ms; likelihood', d['ms_per_likelihood_same_driver'])
PY
done
