cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5f_gradtrace -o kt -- python $R/tools/gradient_bench.py --steps 6 > $R/gpurun_out/r5f_gradtrace.json 2> $R/gpurun_out/r5f_gradtrace.err
find $R/gpurun_out/r5f_gradtrace -name "*.db" -delete 2>/dev/null
f=$(find $R/gpurun_out/r5f_gradtrace -name "*kernel_stats.csv" | head -1)
head -14 $f | cut -c1-200
