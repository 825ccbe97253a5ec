# a model move behind accepted node-height moves (an operation list the engine has not seen): this build against another engine library
# (BEAGLE_MI355_ENGINE_LIB, e.g. a build/variants/<name>/libhmsbeagle-jni.so of tools/build_variant.sh)     bash tools/mixed_chain_ab.sh [other.so]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for L in this ${1:-}; do
  if [ $L = this ]; then unset BEAGLE_MI355_ENGINE_LIB; else export BEAGLE_MI355_ENGINE_LIB=$L; fi
  for P in 0 12500; do A=""; [ $P != 0 ] && A="--patterns $P"
    timeout 300 python bench.py $A --steps 50 --no-cpu-baseline --no-live-traffic --no-library-route 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('partial_update') or {}; print('$L P=$P main ms', d['ms_per_step'], 'median', d['ms_per_step_median'], '| move us', p.get('us_per_branch_move'), '| new list', p.get('full_evaluation_on_a_new_list'))"
  done
done
