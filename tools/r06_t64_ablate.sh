# what bounds k_walkT64 on config C: kernel us per evaluation of builds with parts left out (tools/build_mfma_variant.sh t64_<X> -DMI355_EXP_T64_<X>; wrong results)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in lab NOMFMA NODMA NOLOAD NOSTORE NOLOADSTORE ONLYMFMA; do
  lib=$R/build/variants/t64_$v/libhmsbeagle-jni.so; [ $v = lab ] && lib=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
  echo "$v: $(BEAGLE_MI355_ENGINE_LIB=$lib timeout 120 python tools/walk_time.py 1000000 C 2>&1 | tail -1)"
done
