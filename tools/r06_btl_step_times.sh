for off in 0 1; do
BEAGLE_MI355_NO_SITE_PREFETCH=$off python bench.py --config A --caller btl --steps 200 --warmup 10 --step-times --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['step_ms']
print('off=$off mean', d['ms_per_step'], 'median', d['ms_per_step_median'])
print(' '.join('%.2f'%x for x in s))"
done
