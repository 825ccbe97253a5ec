# A with ALWAYS rescaling (every evaluation in write mode): the wide lane map of write-mode programs against the narrow one
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for V in 0 1; do
  BEAGLE_MI355_NO_WIDE_WRITE=$V timeout 300 python bench.py --rescaling always --steps 100 --no-cpu-baseline --no-live-traffic --no-library-route --no-side-records 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A ALWAYS no_wide=$V', d['value'], 'evals/s kernel us', d['roofline']['kernel_us_per_eval'], 'lnL', d['lnL'])"
done
