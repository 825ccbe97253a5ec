R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for c in "$@"; do
  echo "== $c: $(BEAGLE_MI355_ENGINE_LIB=$R/build/variants/$c/libhmsbeagle-jni.so python tools/r06_flake_diag.py 20 2>&1 | grep -c '^rep') differing evaluations in 8 two-thread repetitions"
done
