# k_walkT64: row tiles of a streamed operand in flight (W64_RING), and the slice size again on the streamed kernel (LAB builds)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { timeout 200 python bench.py --config C --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel_us_per_eval'], r['per_eval']['walks'])"; }
for v in lab t64_ring2 t64_ring6 t64_ring8; do
  lib=$R/build/variants/$v/libhmsbeagle-jni.so; [ $v = lab ] && lib=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
  echo "$v: $(BEAGLE_MI355_ENGINE_LIB=$lib run)"
done
export BEAGLE_MI355_ENGINE_LIB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
for c in 8 12 16 24 32 48 64; do echo "chunk=$c: $(BEAGLE_MI355_CHUNK=$c run)"; done
