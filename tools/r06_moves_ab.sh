# round 6: a product switch's effect on branch moves (partial_update record), alternating: bash tools/r06_moves_ab.sh BEAGLE_MI355_NO_DEEP_PREFETCH
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
VAR=$1
pu() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('partial_update') or {}; print('move us', p.get('us_per_branch_move'), '| new list ms', (p.get('full_evaluation_on_a_new_list') or {}).get('ms_per_full_evaluation_median'), '| chain_mixed', (p.get('chain_mixed') or {}).get('evals_per_s'), (p.get('chain_mixed') or {}).get('us_per_proposal'))"; }
for pass in 1 2; do for t in 0 1; do
  export $VAR=$t
  echo "pass $pass $VAR=$t  12500: $(timeout 200 python bench.py --patterns 12500 --steps 50 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-other-configs 2>/dev/null | pu)"
  echo "pass $pass $VAR=$t  1e5:   $(timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-other-configs 2>/dev/null | pu)"
  echo "pass $pass $VAR=$t  D1:    $(timeout 200 python bench.py --real benchmark1 --steps 50 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-other-configs 2>/dev/null | pu)"
done; done
