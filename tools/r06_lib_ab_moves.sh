# round 6: two engine libraries against each other, full evaluation + branch move + mixed chain on the small alignments: bash tools/r06_lib_ab_moves.sh <a.so> <b.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
pu() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('partial_update') or {}; r=d['roofline']; print(d['value'], 'evals/s, median ms', d.get('ms_per_step_median'), 'kernel', r['kernel_us_per_eval'], 'stored', (r.get('per_eval') or {}).get('stored'), '| move us', p.get('us_per_branch_move'), '| new list ms', (p.get('full_evaluation_on_a_new_list') or {}).get('ms_per_full_evaluation_median'), '| chain_mixed', (p.get('chain_mixed') or {}).get('evals_per_s'), '| lnL', repr(d['lnL']))"; }
for pass in 1 2; do for L in "$@"; do
  export BEAGLE_MI355_ENGINE_LIB=$R/$L
  echo "pass $pass $L"
  echo "   D1:      $(timeout 200 python bench.py --real benchmark1 --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-other-configs 2>/dev/null | pu)"
  echo "   D synth: $(timeout 200 python bench.py --config D --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-other-configs 2>/dev/null | pu)"
done; done
