# round 6: what a node-height move is made of at the metric's size: kernel trace of a short bench run with its side records, the
# dispatches grouped by (kernel, grid) and a window of consecutive ones from the middle of the branch-move section
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/timeline/moves -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --no-other-configs --no-library-route > $R/gpurun_out/timeline/moves.json 2>/dev/null
python - <<PY
import sqlite3, json, collections
db = sqlite3.connect("$R/gpurun_out/timeline/moves/p_results.db")
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
g = collections.OrderedDict()
for r in rows:
    k = (r[0].replace("_ZN5mi355", "")[:40], r[3], r[4])
    a = g.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (r[2] - r[1]) / 1000
for k, (n, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-42s grid %8d x %3d  %6d calls  avg %8.2f us" % (k[0], k[1], k[2], n, t / n))
# the branch moves: walks with a small grid_y; a window from their middle
idx = [i for i, r in enumerate(rows) if "k_walk4_fast" in r[0]]
mid = idx[len(idx) // 2]
t0 = rows[mid - 12][1]
for r in rows[mid - 12: mid + 24]:
    print("  %-46s start %8.1f dur %7.1f grid %d x %d" % (r[0].replace("_ZN5mi355", "")[:46], (r[1] - t0) / 1000, (r[2] - r[1]) / 1000, r[3], r[4]))
d = json.loads(open("$R/gpurun_out/timeline/moves.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("partial_update"))[:900])
PY
