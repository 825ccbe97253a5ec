# round 6: fused cherries against the unfused program (BEAGLE_MI355_NO_CHERRY_FUSION=1), same build, alternating: bash tools/r06_fusion_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; pe=r.get('per_eval') or {}; print(d['value'], 'evals/s  ms/step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), ' kernel us', r['kernel_us_per_eval'], ' fused', pe.get('fused_cherries'), 'of', pe.get('micro_ops'), ' lnL', repr(d['lnL']))"; }
common="--no-cpu-baseline --no-live-traffic --no-library-route --no-side-records"
for pass in 1 2; do for t in 0 1; do
  echo "== pass $pass NO_CHERRY_FUSION=$t"
  export BEAGLE_MI355_NO_CHERRY_FUSION=$t
  echo "A: $(timeout 300 python bench.py --steps 100 --warmup 10 $common 2>/dev/null | line)"
  echo "shard 12500 (sharded path): $(timeout 200 python bench.py --patterns 12500 --force-sharded --steps 200 --warmup 12 $common 2>/dev/null | line)"
  echo "D real1: $(timeout 300 python bench.py --real benchmark1 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "D real2: $(timeout 300 python bench.py --real benchmark2 --steps 300 --warmup 20 $common 2>/dev/null | line)"
  echo "E: $(timeout 300 python bench.py --config E --steps 300 --warmup 20 --no-cpu-baseline --no-live-traffic --no-side-records 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_eval'], d['roofline']['per_eval'].get('fused_cherries'), repr(d['lnL']))")"
done; done
