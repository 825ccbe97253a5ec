#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof_<round><tag>/) into the tracked summaries:

  profiles/<round><tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim copy)
  profiles/<round><tag>_bench.json         the bench line printed by the profiled command
  profiles/<round><tag>_sq_counters.txt    where the hot kernel's wave cycles go (SQ_* pass)
  profiles/hbm_traffic.json                per configuration: HBM bytes per evaluation of the hot kernel — what bench.py
                                           reports as roofline.traffic, keyed to the build by a hash of the kernel sources

HBM bytes from counters (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in units of 1024 B; on gfx950
FETCH_SIZE reports exactly HALF of the bytes of a wide coalesced streaming read (128-B requests tallied at 64 B), so the
read side is DOUBLED; WRITE_SIZE is used as is.  (Round 1 cross-checked both corrections against a kernel whose traffic is
known: 1.0005 x the algorithmic byte count.)  Bytes per evaluation = sum over every dispatch of the hot kernel in the
profiled run / evaluations the run executed (bench line: evaluations_total)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_rows(src, sub, kernel):
    out = []
    for p in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if kernel in r["Kernel_Name"]:
                out.append(r)
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    config = sys.argv[2] if len(sys.argv) > 2 else "A"
    kernel = sys.argv[3] if len(sys.argv) > 3 else "k_walk4"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    out = os.path.join(ROOT, "profiles")
    for p in glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(p, os.path.join(out, tag + "_kernel_stats.csv"))

    def bench_line(name):
        return json.loads(open(os.path.join(src, name)).read().strip().splitlines()[-1])

    bench = bench_line("bench_kt.json")
    json.dump(bench, open(os.path.join(out, tag + "_bench.json"), "w"), indent=1)
    rec = {"round": tag, "kernel": kernel, "kernel_source_hash": bench.get("kernel_source_hash")}
    for sub, name, factor in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
        rows = [r for r in counter_rows(src, sub, kernel) if r["Counter_Name"] == name]
        b = bench_line("bench_%s.json" % sub)
        evals = max(1, int(b.get("evaluations_total", 1)))
        total = sum(float(r["Counter_Value"]) for r in rows) * 1024.0 * factor
        rec["%s_bytes_per_eval" % ("read" if sub == "fetch" else "write")] = int(total / evals)
        rec["%s_dispatches" % sub] = len(rows)
        rec["%s_evaluations" % sub] = evals
    rec["bytes_per_eval"] = rec["read_bytes_per_eval"] + rec["write_bytes_per_eval"]
    rec["design_bytes_per_eval"] = bench["roofline"].get("bytes_per_eval")
    rec["algorithmic_bytes_per_eval"] = bench["roofline"].get("algorithmic_bytes_per_eval")
    rec["kernel_us_per_eval_unprofiled"] = bench["roofline"].get("kernel_us_per_eval")
    rec["HBM_GBps_from_counters"] = round(rec["bytes_per_eval"] / (bench["roofline"]["kernel_us_per_eval"] * 1e-6) / 1e9, 1)
    rec["corrections"] = "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B), x1024 B per unit; WRITE_SIZE x1024"
    rec["source"] = "gpurun_out/prof_%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)" % tag
    path = os.path.join(out, "hbm_traffic.json")
    allrec = json.load(open(path)) if os.path.exists(path) else {}
    if "kernel" in allrec:          # round-1 layout (one flat record): start over
        allrec = {}
    allrec[config] = rec
    json.dump(allrec, open(path, "w"), indent=1)
    # every kernel of the run, not only the hot one: HBM bytes per evaluation by kernel (same corrections), so that what the
    # configuration moves beyond its hot kernel — gathers, snapshots, matrices, root — has a name
    by = {}
    evals_of = {}
    for sub, name, factor in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
        evals_of[sub] = max(1, int(bench_line("bench_%s.json" % sub).get("evaluations_total", 1)))
        for r in counter_rows(src, sub, ""):
            if r["Counter_Name"] != name:
                continue
            kn = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mi355::", "")[:44]
            e = by.setdefault(kn, {"fetch": 0.0, "write": 0.0, "n": 0})
            e[sub] += float(r["Counter_Value"]) * 1024.0 * factor
            e["n"] += 1 if sub == "fetch" else 0
    if by:
        with open(os.path.join(out, tag + "_traffic_by_kernel.txt"), "w") as fh:
            fh.write("# HBM bytes per evaluation by kernel, config %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH x2, x1024 B per unit);\n"
                     "# %d / %d evaluations in the two passes, set-up launches included (divided over the same evaluations)\n" % (config, evals_of["fetch"], evals_of["write"]))
            fh.write("%-46s %10s %14s %14s\n" % ("kernel", "launches", "read MB/eval", "written MB/eval"))
            for kn, e in sorted(by.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"])):
                fh.write("%-46s %10d %14.2f %14.2f\n" % (kn, e["n"], e["fetch"] / evals_of["fetch"] / 1e6, e["write"] / evals_of["write"] / 1e6))
    # SQ pass
    agg = {}
    for r in counter_rows(src, "sq", kernel):
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if agg:
        with open(os.path.join(out, tag + "_sq_counters.txt"), "w") as fh:
            fh.write("# rocprofv3 --pmc SQ_* -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline (config %s); sums over the dispatches of %s\n" % (config, kernel))
            wc = sum(agg.get("SQ_WAVE_CYCLES", [0])) or 1.0
            for k, v in sorted(agg.items()):
                fh.write("%-24s %.4g   (%.3f of SQ_WAVE_CYCLES)\n" % (k, sum(v), sum(v) / wc))
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
