# round 6: kernel timeline (start offsets and durations, us) of the last dispatches of ANY command: bash tools/r06_timeline_cmd.sh <name> <rows> <command...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; n=$1; rows=$2; shift; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/timeline/$n -o p -- "$@" > /dev/null 2>&1
python - <<PY
import sqlite3
db = sqlite3.connect("$R/gpurun_out/timeline/$n/p_results.db")
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
t0 = rows[-$rows][1]
for r in rows[-$rows:]:
    print("  %-48s start %8.1f dur %7.1f grid %d x %d" % (r[0].replace("_ZN5mi355", "")[:48], (r[1] - t0) / 1000, (r[2] - r[1]) / 1000, r[3], r[4]))
PY
