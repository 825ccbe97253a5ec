import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.split("(")[0].replace("mi355::", "")[:50]
idx = [i for i, r in enumerate(rows) if "foldRecip" in r[2]]
print(len(rows), "kernels; fold launches at", idx)
if idx:
    c = idx[int(sys.argv[2])] if len(sys.argv) > 2 else idx[-1]
    lo = max(0, c - 14); hi = min(len(rows), c + 16)
    t0 = rows[lo][0]; pe = rows[lo][0]
    for s, e, n in rows[lo:hi]:
        print("%9.1f us  +gap %7.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - pe) / 1e3, (e - s) / 1e3, short(n)))
        pe = e
