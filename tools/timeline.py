"""Per-evaluation kernel timeline from a rocprofv3 --kernel-trace CSV: for the last evaluations of the run, every kernel in
stream order with its duration and the gap to the previous one.  python tools/timeline.py <dir>"""
import csv
import glob
import os
import sys

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    raise SystemExit("no kernel_trace.csv under %s" % d)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# an evaluation ends with the root reduction
ends = [i for i, r in enumerate(rows) if "rootFinal" in r[2] or "k_rootLogLikelihood" in r[2] or "k_rootSite" in r[2]]
short = lambda n: n.split("(")[0].replace("mi355::", "")[:60]
if len(ends) < 4:
    raise SystemExit("too few evaluations in the trace")
last = ends[-1]
first = ends[-3] + 1
prev_end = rows[first - 1][1]
t0 = rows[first][0]
for s, e, n in rows[first:last + 1]:
    print("%9.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short(n)))
    prev_end = e
print("span of the last two evaluations: %.1f us" % ((rows[last][1] - t0) / 1e3))
