#!/bin/bash
# round 6: BeagleTreeLikelihood's protocol (site values read back every evaluation) with and without the site prefetch, alternating on one box
#   bash tools/r06_site_prefetch_ab.sh [config] [extra bench.py arguments]
cfg=${1:-A}; shift
for rep in 1 2; do
  for off in 0 1; do
    BEAGLE_MI355_NO_SITE_PREFETCH=$off python bench.py --config $cfg --caller btl --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" 2>/dev/null | tail -1 |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NO_SITE_PREFETCH=$off', d['config'].get('caller'), d['value'], 'evals/s', d['ms_per_step'], 'ms/step, median', d.get('ms_per_step_median'), '| other caller', d.get('other_caller'))"
  done
done
