# config C (61 states) on the walk without hold slots: slice size sweep + the plan of the default (LAB build): bash tools/r06_t64_sweep.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export BEAGLE_MI355_ENGINE_LIB=$R/beast-mcmc_amd/lib/lab/libhmsbeagle-jni.so
run() { timeout 200 python bench.py --config C --steps 20 --warmup 4 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records --no-other-configs 2>/tmp/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel_us_per_eval'], r['launches_per_eval'], r['per_eval']['stored'], r['per_eval']['mem_reads'], r['per_eval']['walks'])"; }
echo "default: $(BEAGLE_MI355_DUMP_PLAN=1 run)"; grep "plan:" /tmp/err.txt | sort | uniq -c | head -5
for c in 0 12 16 24 32 48 64 100; do echo "chunk=$c: $(BEAGLE_MI355_CHUNK=$c BEAGLE_MI355_DUMP_PLAN=1 run)"; grep "plan:" /tmp/err.txt | sort | uniq -c | sort -rn | head -2 | cut -c1-400; done
echo "levels (NO_T64_WALK): $(BEAGLE_MI355_NO_T64_WALK=1 run)"
