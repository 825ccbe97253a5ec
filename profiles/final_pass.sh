#!/bin/bash
# The round's final measurement pass in ONE GPU-box call:   bash profiles/final_pass.sh r06
#   1. the GPU test tier (log kept)
#   2. the bench lines of configs A-E as the driver runs them (CPU baseline, roofline.traffic measured live by bench.py itself,
#      the in-library multi-GPU route appended), plus the protocol variants: BeagleTreeLikelihood caller, ALWAYS rescaling, the
#      12 500-pattern shard, config D on the reference's two real benchmark alignments
#   3. rocprofv3 passes of profiles/collect.sh for A, B, C and E (kernel stats, FETCH/WRITE, SQ counters) and their summaries
#      (profiles/summarize.py: <round>_<cfg>_kernel_stats.csv, _sq_counters.txt, hbm_traffic.json keyed to this build)
# Everything the repo tracks of it is copied to gpurun_out/profiles_final/ (the box's profiles/ does not travel back).
R=${1:-r05}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/profiles_final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/profiles_final/${R}_pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/profiles_final/${R}_pytest_gpu.log)"
line() { python -c "import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);r=d['roofline'];print(sys.argv[2], d['value'],'evals/s kernel us',r['kernel_us_per_eval'],'frac',r['frac'],'traffic',r['traffic'],'|',r['traffic_source'][:60],'| cpu',d['cpu_baseline'] and d['cpu_baseline']['value'],'| lib',d.get('library_route') and d['library_route'].get('value'))" "$1" "$2" 2>&1 | tail -1; }
for cfg in A B C D E; do
  steps=200; [ $cfg = B ] && steps=60; [ $cfg = C ] && steps=60
  timeout 600 python bench.py --config $cfg --steps $steps --warmup 5 2> gpurun_out/${R}_bench_$cfg.err | tail -1 > gpurun_out/profiles_final/${R}_bench_$cfg.json
  line gpurun_out/profiles_final/${R}_bench_$cfg.json "bench $cfg"
done
# the driver's own command line (20 timed steps after 5 warm-up steps): shard_point, partial_update, library_route on it
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${R}_bench_A_driver.err | tail -1 > gpurun_out/profiles_final/${R}_bench_A_driver_cmdline.json
line gpurun_out/profiles_final/${R}_bench_A_driver_cmdline.json "A as the driver runs it"
python -c "import json;d=json.loads(open('gpurun_out/profiles_final/${R}_bench_A_driver_cmdline.json').read());print('  shard_point', d.get('shard_point'));print('  partial_update', d.get('partial_update'));print('  library_route', d.get('library_route'))"
# one GPU's share of a 2- / 4- / 8-GPU job through the multi-GPU code path (engine-side ncclAllReduce, communicator of one rank)
for p in 50000 25000 12500; do
  timeout 300 python bench.py --patterns $p --force-sharded --steps 200 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_bench_A_shard${p}_sharded_path.json; line gpurun_out/profiles_final/${R}_bench_A_shard${p}_sharded_path.json "A shard $p (sharded path)"
done
# gradients (secondary): 4 states at 20 000 and 1e5 patterns, 20 and 61 states at their configs' sizes
timeout 300 python tools/gradient_bench.py --config A --patterns 20000 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_gradient_bench.json
timeout 300 python tools/gradient_bench.py --config A --steps 5 --warmup 3 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_gradient_bench_1e5.json
timeout 600 python tools/gradient_bench.py --config B --steps 5 --warmup 3 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_gradient_bench_B.json
timeout 600 python tools/gradient_bench.py --config C --steps 5 --warmup 3 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_gradient_bench_C.json
for g in "" _1e5 _B _C; do python -c "import json;d=json.loads(open('gpurun_out/profiles_final/${R}_gradient_bench$g.json').read());print('gradient$g', d['ms_per_gradient'],'ms, likelihood', d['ms_per_likelihood_same_driver'],'ms, roofline frac', d['roofline']['frac'], d['how'])" 2>&1 | tail -1; done
timeout 300 python bench.py --config A --caller btl --steps 100 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_bench_A_btl.json; line gpurun_out/profiles_final/${R}_bench_A_btl.json "A btl"
# BeagleTreeLikelihood's protocol with and without the site prefetch (round 6), alternating; and the step times of a long DYNAMIC chain
(bash tools/r06_site_prefetch_ab.sh A; bash tools/r06_site_prefetch_ab.sh D --real benchmark1) > gpurun_out/profiles_final/${R}_site_prefetch_ab.txt 2>&1; cat gpurun_out/profiles_final/${R}_site_prefetch_ab.txt
bash tools/r06_cycle_times.sh 2>&1 | grep -v "gather launch\|amdgpu.ids" > gpurun_out/profiles_final/${R}_cycle_times_A.txt; head -18 gpurun_out/profiles_final/${R}_cycle_times_A.txt
timeout 300 python bench.py --config A --rescaling always --steps 100 --warmup 5 --no-cpu-baseline --no-library-route 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_bench_A_always.json; line gpurun_out/profiles_final/${R}_bench_A_always.json "A always"
timeout 300 python bench.py --config B --rescaling always --steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic --no-library-route 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_bench_B_always.json; line gpurun_out/profiles_final/${R}_bench_B_always.json "B always"
timeout 300 python bench.py --patterns 12500 --steps 200 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_bench_A_shard12500.json; line gpurun_out/profiles_final/${R}_bench_A_shard12500.json "A shard"
python -c "import json;d=json.loads(open('gpurun_out/profiles_final/${R}_bench_A_shard12500.json').read());print('  partial_update at 12 500 patterns', d.get('partial_update'))"
timeout 300 python tools/readback_bench.py 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_readback.json; echo "readback $(cut -c1-300 gpurun_out/profiles_final/${R}_readback.json)"
timeout 300 python tools/gradient_bench.py --config A --steps 5 --warmup 3 --rescale 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_gradient_bench_1e5_rescale.json
BEAGLE_MI355_GRADIENT_VIRTUAL=2 timeout 300 python tools/gradient_bench.py --config A --steps 5 --warmup 3 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_gradient_bench_1e5_unstored_definitions.json
BEAGLE_MI355_GRADIENT_VIRTUAL=0 timeout 300 python tools/gradient_bench.py --config A --steps 5 --warmup 3 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_gradient_bench_1e5_every_node_stored.json
for g in _1e5_rescale _1e5_unstored_definitions _1e5_every_node_stored; do python -c "import json;d=json.loads(open('gpurun_out/profiles_final/${R}_gradient_bench$g.json').read());print('gradient$g', d['ms_per_gradient'],'ms, likelihood', d['ms_per_likelihood_same_driver'],'ms, stored', d.get('post_order_nodes_stored_per_gradient'), d['how'])" 2>&1 | tail -1; done
BEAGLE_MI355_NO_SCALE_FOLD=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-library-route --no-side-records 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_bench_A_driver_cmdline_no_fold.json; line gpurun_out/profiles_final/${R}_bench_A_driver_cmdline_no_fold.json "A, driver command line, per-node factors (NO_SCALE_FOLD)"
# the gradient chain at 1e5 patterns: kernel trace and the pre-order walk's SQ counters
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${R}_grad/kt -o kt -- python $ROOT/tools/gradient_bench.py --config A --steps 4 --warmup 3 > /dev/null 2>&1)
cp $(find gpurun_out/prof_${R}_grad/kt -name "*kernel_stats.csv" | head -1) gpurun_out/profiles_final/${R}_gradient_1e5_kernel_stats.csv 2>/dev/null
KERNEL=preWalk4 CMD="python $ROOT/tools/gradient_bench.py --config A --steps 2 --warmup 3" bash tools/prof_counters.sh ${R}_gradsq 2>&1 | grep -v "rc=0" > gpurun_out/profiles_final/${R}_gradient_prewalk_sq_counters.txt
for real in benchmark1 benchmark2; do
  timeout 300 python bench.py --real $real --steps 200 --warmup 5 --no-live-traffic --no-library-route 2>/dev/null | tail -1 > gpurun_out/profiles_final/${R}_bench_D_$real.json; line gpurun_out/profiles_final/${R}_bench_D_$real.json "D $real"
done
TAG=_shard STEPS=20 bash profiles/collect.sh $R --config A --patterns 12500 2>&1 | grep rc=
python profiles/summarize.py ${R}_shard A k_walk4 > /dev/null          # (hbm_traffic.json's "A" entry is overwritten by the full size below)
python tools/timeline.py gpurun_out/prof_${R}_shard/kt > gpurun_out/profiles_final/${R}_shard_timeline.txt 2>&1
TAG=_A STEPS=10 bash profiles/collect.sh $R --config A 2>&1 | grep rc=
TAG=_B STEPS=5 bash profiles/collect.sh $R --config B 2>&1 | grep rc=
TAG=_C STEPS=5 bash profiles/collect.sh $R --config C 2>&1 | grep rc=
TAG=_E STEPS=20 bash profiles/collect.sh $R --config E 2>&1 | grep rc=
TAG=_B_always STEPS=5 bash profiles/collect.sh $R --config B --rescaling always 2>&1 | grep rc=     # (round 6: the write-mode T32 walk, k_walkT32W1)
python profiles/summarize.py ${R}_A A k_walk4 > /dev/null; python profiles/summarize.py ${R}_B B k_walkT32 > /dev/null
python profiles/summarize.py ${R}_C C k_walkT64 > /dev/null; python profiles/summarize.py ${R}_E E k_walk4 > /dev/null
python profiles/summarize.py ${R}_B_always B_always k_walkT32W1 > /dev/null
cp profiles/hbm_traffic.json profiles/${R}_*_kernel_stats.csv profiles/${R}_*_sq_counters.txt profiles/${R}_*_traffic_by_kernel.txt profiles/${R}_?_bench.json profiles/${R}_B_always_bench.json profiles/${R}_shard_bench.json gpurun_out/profiles_final/ 2>/dev/null
python -c "import json;d=json.load(open('profiles/hbm_traffic.json'));[print(k, v['bytes_per_eval'], v.get('design_bytes_per_eval'), v.get('HBM_GBps_from_counters')) for k,v in d.items()]"
ls gpurun_out/profiles_final | wc -l
