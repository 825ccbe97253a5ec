#!/bin/bash
# round 6: the wall-clock time of every step of a long DYNAMIC chain (the rescaling evaluations of every 100th step stand out)
#   bash tools/r06_cycle_times.sh [bench.py arguments]
BEAGLE_MI355_HOST_TIMING=2 python bench.py --steps 520 --warmup 10 --step-times --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records "$@" 2> /tmp/cycle.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['step_ms']; import statistics as st
med=st.median(s); print('mean', d['ms_per_step'], 'median', round(med,4), 'steps', len(s))
for i,x in enumerate(s):
    if x > 1.25*med: print('  step', i, x)
print('sum over median of the slow steps, ms per 100 steps:', round(sum(x-med for x in s if x>1.25*med)/len(s)*100,3))"
grep -v "^\[W\|Warning" /tmp/cycle.err | tail -40
