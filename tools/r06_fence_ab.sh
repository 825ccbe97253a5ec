# ticket hand-over fences (kernels_walk4.hip MI355_TICKET_FENCE: 1 consumer buffer_inv sc1, 2 producer buffer_wbl2 sc1, 3 both, 4 a delay):
# failures of the two-thread test with the XCD map off (fails 10 / 10 without a fence), then config A's rate
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
N=${1:-10}
for v in 1 2 3 4; do
  lib=$R/build/variants/fence$v/libhmsbeagle-jni.so; f=0
  for i in $(seq 1 $N); do
    BEAGLE_MI355_ENGINE_LIB=$lib BEAGLE_MI355_NO_XCD_MAP=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "concurrent_instances" 2>&1 | tail -1 | grep -q failed && f=$((f+1))
  done
  echo "fence $v: $f / $N failed (XCD map off)"
done
for v in 0 1 2 3; do
  lib=$R/build/variants/fence$v/libhmsbeagle-jni.so; [ $v = 0 ] && lib=$R/beast-mcmc_amd/lib/libhmsbeagle-jni.so
  echo "fence $v config A: $(BEAGLE_MI355_ENGINE_LIB=$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us_per_eval'])")"
  echo "fence $v shard 12500: $(BEAGLE_MI355_ENGINE_LIB=$lib python bench.py --steps 200 --warmup 10 --patterns 12500 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us_per_eval'])")"
done
