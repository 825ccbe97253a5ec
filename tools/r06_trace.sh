# LAB build: workgroup trace of the 12 500-pattern shard's launch (BEAGLE_MI355_WALK_TRACE) under a few settings
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export BEAGLE_MI355_ENGINE_LIB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
P=${P:-12500}
run() { echo "== $*"; env "$@" BEAGLE_MI355_DUMP_PLAN=1 BEAGLE_MI355_WALK_TRACE=30 timeout 150 python bench.py --steps 40 --patterns $P --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>&1 | grep "mi355" | grep -v "plan:" | head -60; env "$@" BEAGLE_MI355_DUMP_PLAN=1 timeout 150 python bench.py --steps 3 --warmup 1 --patterns $P --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>&1 | grep "plan:" | tail -1 | cut -c1-900; }
run BEAGLE_MI355_NO_WALK_TICKETS=0
run BEAGLE_MI355_NO_WALK_TICKETS=1
run BEAGLE_MI355_NO_WALK_TICKETS=0 BEAGLE_MI355_CHUNK=128 BEAGLE_MI355_CHUNK_TOP=8
