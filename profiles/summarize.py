#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof_<round>/) into the tracked summaries:

  profiles/<round>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (verbatim copy)
  profiles/<round>_prune_launches.csv    one row per pruning launch of the last evaluation: ops in the launch, duration,
                                         HBM read / write bytes from the two PMC passes
  profiles/<round>_bench.json            the bench line printed by the profiled command
  profiles/hbm_traffic.json              what bench.py reports as roofline.traffic (bytes per launch of the dominant kernel)

HBM bytes from counters (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1024 B;
on gfx950 FETCH_SIZE reports exactly HALF of the bytes of a wide coalesced streaming read (128-B requests tallied
at 64 B), so the read side is DOUBLED; WRITE_SIZE is used as is.  Both corrections are cross-checked below against the
algorithmic byte count of the same launches (they agree to <1 %, which also shows the Infinity Cache absorbs nothing
at this working-set size).
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    kernel = sys.argv[2] if len(sys.argv) > 2 else "k_prune4"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + rnd)
    out = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(out, rnd + "_kernel_stats.csv"))
    bench = json.loads(open(os.path.join(src, "bench_kt.json")).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(out, rnd + "_bench.json"), "w"), indent=1)
    per_eval = int(round(bench["roofline"]["launches_per_eval"]))

    trace = [r for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_trace.csv"))) if kernel in r["Kernel_Name"]]

    def pmc(sub, name):
        path = os.path.join(src, sub, sub + "_counter_collection.csv")
        return [r for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]

    fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
    last_t, last_f, last_w = trace[-per_eval:], fetch[-per_eval:], write[-per_eval:]
    rows = []
    for t, f, w in zip(last_t, last_f, last_w):
        dur = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
        rd = 2.0 * float(f["Counter_Value"]) * 1024.0
        wr = float(w["Counter_Value"]) * 1024.0
        rows.append({"ops_in_launch": int(t["Grid_Size_Y"]), "duration_us": round(dur, 2),
                     "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr),
                     "GBps": round((rd + wr) / dur / 1e3, 1), "vgpr": t["VGPR_Count"], "lds_bytes": t["LDS_Block_Size"]})
    with open(os.path.join(out, rnd + "_prune_launches.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    rd = sum(r["hbm_read_bytes"] for r in rows)
    wr = sum(r["hbm_write_bytes"] for r in rows)
    dur = sum(r["duration_us"] for r in rows)
    alg = bench["roofline"]["algorithmic_bytes_per_launch"] * per_eval
    summary = {
        "round": rnd, "kernel": kernel, "launches_per_eval": per_eval,
        "bytes_per_launch": int((rd + wr) / per_eval),
        "read_bytes_per_eval": rd, "write_bytes_per_eval": wr, "algorithmic_bytes_per_eval": alg,
        "traffic_over_algorithmic": round((rd + wr) / alg, 4),
        "kernel_time_us_per_eval": round(dur, 1), "HBM_GBps_from_counters": round((rd + wr) / dur / 1e3, 1),
        "corrections": "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B), x1024 B per unit; WRITE_SIZE x1024",
        "source": "gpurun_out/prof_%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)" % rnd,
    }
    json.dump(summary, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(out, rnd + "_hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))
    avg = sum((int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) for t in trace) / len(trace) / 1e3
    print("kernel-trace average %s launch: %.2f us over %d launches; bench.py HIP-event average: %.2f us"
          % (kernel, avg, len(trace), bench["roofline"]["avg_launch_us"]))


if __name__ == "__main__":
    main()
