#!/bin/bash
# The round's final measurement pass in ONE GPU-box call: GPU test tier, bench lines + rocprofv3 passes (run_round.sh),
# summaries (so that roofline.traffic matches this build), then the bench lines that carry the counters' traffic.
# Everything the repo tracks of it is copied to gpurun_out/profiles_final/ (the box's profiles/ does not travel back).
R=${1:-r02}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash profiles/run_round.sh $R 2>&1 | grep -E "^bench|rc="
python profiles/summarize.py ${R}_A A k_walk4 > /dev/null; python profiles/summarize.py ${R}_B B k_pruneTiled > /dev/null; python profiles/summarize.py ${R}_C C k_pruneTiled > /dev/null
for cfg in A B C; do
  steps=200; [ $cfg != A ] && steps=60
  timeout 300 python bench.py --config $cfg --steps $steps --warmup 5 2>/dev/null | tail -1 > profiles/${R}_bench_$cfg.json
  python -c "import json;d=json.loads(open('profiles/${R}_bench_$cfg.json').read());print('$cfg', d['value'], d['roofline']['frac'], d['roofline']['traffic'])"
done
cp gpurun_out/${R}_bench_D.json gpurun_out/${R}_bench_E.json profiles/
timeout 120 python bench.py --config A --caller btl --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${R}_bench_A_btl.json
timeout 120 python bench.py --patterns 12500 --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 > profiles/${R}_bench_A_shard12500.json
mkdir -p gpurun_out/profiles_final; cp profiles/hbm_traffic.json profiles/${R}_* gpurun_out/profiles_final/
