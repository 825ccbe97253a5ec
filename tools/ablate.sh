cd $GRAFT_REPO_ROOT
for a in ${ABL:-0 16 24 256 512 1024 2048 4096 2304}; do
  BEAGLE_MI355_ABLATE=$a timeout 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate=$a', 'evals/s', d['value'], 'walk_us', d['roofline']['avg_launch_us'], 'lnL', d['lnL'])"
done
