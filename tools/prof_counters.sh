# SQ counter passes over the headline bench:  bash tools/prof_counters.sh <tag> [bench.py arguments]   (KERNEL=<substring of the kernel name>, default walk4;
# CMD="python tools/gradient_bench.py --patterns 100000 --steps 2" profiles that command instead of bench.py)
# (run on the GPU box through gpurun; never add TA_* _sum counters: that pass hangs rocprofv3)
ROOT=$GRAFT_REPO_ROOT; TAG=${1:-sq}; shift; ARGS="$@"; OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- ${CMD:-python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-library-route} $ARGS > $OUT/bench_$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run sqa SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
run sqb SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
run sqc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES SQ_INSTS_VMEM_WR
run sqd SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES
find $OUT -name "*.db" -delete
python3 $ROOT/tools/sum_counters.py $OUT ${KERNEL:-walk4}
