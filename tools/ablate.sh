cd $GRAFT_REPO_ROOT
for env in "BEAGLE_MI355_CHUNK=0" "BEAGLE_MI355_CHUNK=24" "BEAGLE_MI355_CHUNK=64" "BEAGLE_MI355_CHUNK=150" "BEAGLE_MI355_CHUNK=300"; do
 for args in "" "--patterns 50000"; do
  env $env timeout 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env args=[$args]', 'evals/s', d['value'], 'ms/step', d['ms_per_step'], 'kernel_us/eval', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_eval'],1), 'lnL', d['lnL'])"
 done
done
