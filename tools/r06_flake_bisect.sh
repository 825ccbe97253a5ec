# which switch makes test_concurrent_instances_from_two_threads stable: failures out of N runs per switch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
N=${1:-10}
for sw in NONE NO_LAUNCH_FUSION NO_ROOT_FUSION NO_WALK_TICKETS NO_WALK_FUSION NO_CHERRY_FUSION NO_LOAD_SKIP NO_SLICE_SUMS NO_SCALE_FOLD NO_PLAN_CACHE NO_T32_WALK NO_T32_WRITE_WALK NO_FAST_WALK NO_VIRTUAL NO_XCD_MAP; do
  f=0
  for i in $(seq 1 $N); do
    env BEAGLE_MI355_$sw=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "concurrent_instances" 2>&1 | tail -1 | grep -q failed && f=$((f+1))
  done
  echo "$sw: $f / $N failed"
done
