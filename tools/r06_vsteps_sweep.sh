# round 6, LAB build: the cap on a virtual definition (BEAGLE_MI355_VSTEPS) for alignments of few pattern groups: full evaluation, branch move, mixed chain
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export BEAGLE_MI355_ENGINE_LIB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
pu() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('partial_update') or {}; r=d['roofline']; print('eval median ms', d.get('ms_per_step_median'), 'kernel', r['kernel_us_per_eval'], 'stored', (r.get('per_eval') or {}).get('stored'), '| move us', p.get('us_per_branch_move'), '| new list ms', (p.get('full_evaluation_on_a_new_list') or {}).get('ms_per_full_evaluation_median'), '| chain_mixed', (p.get('chain_mixed') or {}).get('evals_per_s'))"; }
for v in default 1 2 4 8 16; do
  if [ $v = default ]; then unset BEAGLE_MI355_VSTEPS; else export BEAGLE_MI355_VSTEPS=$v; fi
  echo "VSTEPS=$v  D1:    $(timeout 200 python bench.py --real benchmark1 --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-other-configs 2>/dev/null | pu)"
  echo "VSTEPS=$v  D2:    $(timeout 200 python bench.py --real benchmark2 --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-other-configs 2>/dev/null | pu)"
  echo "VSTEPS=$v  12500: $(timeout 200 python bench.py --patterns 12500 --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-other-configs 2>/dev/null | pu)"
done
