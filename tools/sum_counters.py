"""Sum rocprofv3 --pmc counters over the dispatches of the walk kernel: python tools/sum_counters.py <dir with sqa/ sqb/ sqc/> [kernel name substring]."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "walk4"
for d in ["sqa", "sqb", "sqc", "sqd"]:
    for f in glob.glob(root + "/" + d + "/**/*counter_collection.csv", recursive=True):
        tot, n = collections.defaultdict(float), 0
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                n += 1
        print(d, n, "counter rows")
        for k, v in sorted(tot.items()):
            print("  %-32s %.4g" % (k, v))
