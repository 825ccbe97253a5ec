#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's `roofline` object.  Run ON THE GPU BOX from the repo root:
#     bash profiles/collect.sh r02 [bench.py arguments, e.g. --config B]
# Separate passes (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass, and
# counters are never combined with other trace domains):
#   1. --kernel-trace --stats   per-kernel average duration over the same command as the bench line
#   2. --pmc FETCH_SIZE         HBM/fabric read traffic per dispatch
#   3. --pmc WRITE_SIZE         HBM/fabric write traffic per dispatch
#   4. --pmc SQ_*               where the waves' cycles go (parked / issue-stalled / issuing)
# Raw output lands in gpurun_out/prof_<round><tag>/ (scratch); profiles/summarize.py turns it into the tracked
# summaries under profiles/.
R=${1:-r02}; shift
TAG=${TAG:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$R$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-20}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- \
    python "$ROOT/bench.py" --steps $STEPS --warmup 3 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > "$OUT/bench_kt.json" 2> "$OUT/kt.err"
echo "kernel-trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- \
    python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
echo "pmc FETCH_SIZE rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- \
    python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > "$OUT/bench_write.json" 2> "$OUT/write.err"
echo "pmc WRITE_SIZE rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES \
    --output-format csv -d "$OUT/sq" -o sq -- \
    python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" > "$OUT/bench_sq.json" 2> "$OUT/sq.err"
echo "pmc SQ rc=$?"
# keep the merge small: the per-dispatch CSVs are all that the summaries need
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"
