#!/usr/bin/env python3
"""Headline benchmark: full-tree log-likelihood evaluations per second, GTR+G4, 1000 taxa x 1e5 unique
patterns, fp64 (BASELINE.json `metric`, config "GTR+G4 nucleotide (4-state), 1000 taxa x 1e5 unique patterns").

One "step" = one full-tree evaluation driven through the C ABI exactly as BEAST drives it after a substitution-model
+ site-model move: setEigenDecomposition + setCategoryRates (NEW values every step: the parameters cycle through THREE
nearby models while the caller flips between two eigen slots, so every slot receives values it does not hold and nothing the
engine could skip as "unchanged" is — perturbed_models) + updateTransitionMatrices(2T-2 branches) +
updatePartials(T-1 level-ordered ops, steady-state DYNAMIC rescaling: read mode) + setCategoryWeights /
setStateFrequencies + calculateRootLogLikelihoods, with the scalar result read back on the host (SURVEY 8d "Timing
protocol").  `--caller btl` adds what the class north_star names, BeagleTreeLikelihood, does on top of that every
evaluation: getSiteLogLikelihoods (P doubles back to the host, BeagleTreeLikelihood.java:1050); the default is the
TreeDataLikelihood protocol (BeagleDataLikelihoodDelegate.java:904-935), and the line reports both.
All inputs are resident in HBM before the timed region.

--config A (default) | B (20 states, 500 x 5e4) | C (61 states, 200 x 2e4) | D (benchmark1-like) | E (Makona-like, four
partitions on one instance through updatePartialsByPartition).

N > 1: one rank per GPU.  Started under torch.distributed.run (what the driver does) the ranks are there already; started
PLAINLY (`python bench.py --gpus N`) this script brings them up itself (re-executes under `python -m torch.distributed.run
--nproc-per-node N --master-addr 127.0.0.1`), and it refuses to print a line whose n_gpus differs from --gpus.  The patterns
(of every partition) are split into contiguous blocks (Patterns.java:142-167), every rank evaluates its block, ONE RCCL
all-reduce of the per-shard lnL per step (E: of the partitionCount per-partition values).  Total work is fixed as N grows
-> "scaling": "strong".  Reference equivalent: -beagle_instances N -beagle_order 1,..,N
(src/dr/evomodelxml/treedatalikelihood/TreeDataLikelihoodParser.java:205-278).
`--route library` times the OTHER multi-GPU route instead: ONE process, the library's resource G+1 ("all GPUs",
csrc/sharded.cpp: one engine instance per GPU on its own host thread, ncclAllReduce of the root sums); with the default
`--route ranks` rank 0 appends a shorter run of that route to the line as "library_route" (configs A-D).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TIMER_EVERY = 4            # the engine's HIP-event kernel timer brackets every 4th evaluation of a timed region (bench_single)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable with a copy kernel)
KERNEL_SOURCES = ("kernels_walk4.hip", "walk4_fast_loop.inc", "kernels_mfma.hip", "kernels.hip", "planner.cpp", "engine_walk.cpp", "engine_levels.cpp", "engine_instance.cpp")


def kernel_source_hash():
    """Identifies the build a profile belongs to: sha256 over the kernel and planner sources."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "beast-mcmc_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def prune_bytes_per_eval(wl):
    """ALGORITHMIC HBM bytes of the pruning of ONE full-tree evaluation: per op the destination write B + B
    per internal child (P bytes per compact tip child) + one 8-byte scale factor per pattern; B = P*S*C*8."""
    t, p = wl.tip_count, wl.pattern_count
    b = p * wl.state_count * wl.category_count * 8
    total = 0
    tree = wl.tree
    for n in range(t, 2 * t - 1):
        total += b + 8 * p
        for ch in (int(tree.left[n]), int(tree.right[n])):
            total += p if ch < t else b
    return total


def perturbed_models(bm, wl, config):
    """THREE nearby substitution + site models per workload; a step takes the next one, as a chain's substitution- and
    site-model moves would.  Three, because the caller flips between two eigen slots every step (BufferIndexHelper): with two
    models each slot would receive the same values every time and the engine's unchanged-upload check (engine_abi.cpp
    uploadIfChanged) would skip that upload; with three every slot gets values it did not hold before, every step."""
    import numpy as np
    from beast_mcmc_amd.inputs import substmodel
    from beast_mcmc_amd.inputs.siterates import GammaSiteRateModel
    out = [(wl.eig, wl.freqs, wl.cat_rates, wl.cat_weights)]
    pi = np.asarray(wl.freqs)
    if config == "A":
        eig2 = substmodel.gtr([1.0, 4.0 * (1 + 1e-6), 0.8, 1.2, 4.5, 1.0], pi)
    elif config in ("D",) and wl.name.startswith("benchmark2"):
        eig2 = substmodel.gtr([1.0, 1.0 + 1e-6, 1.0, 1.0, 1.0, 1.0], pi)
    elif config in ("D",):
        eig2 = substmodel.hky(2.0 * (1 + 1e-6), pi)
    else:                                   # B, C: the same rate matrix at a slightly different normalisation
        eig2 = substmodel.EigenDecomposition(wl.eig.evec, wl.eig.ievc, np.asarray(wl.eig.evals) * (1 + 1e-6))
    if wl.category_count > 1:
        r2, w2 = GammaSiteRateModel(alpha=0.5 * (1 + 1e-6), gamma_categories=wl.category_count).category_rates_and_proportions()
    else:
        r2, w2 = wl.cat_rates, wl.cat_weights
    out.append((eig2, wl.freqs, np.asarray(r2, dtype=float), np.asarray(w2, dtype=float)))
    # the third: the first model's rate matrix at a slightly different normalisation, a third shape parameter
    eig3 = substmodel.EigenDecomposition(wl.eig.evec, wl.eig.ievc, np.asarray(wl.eig.evals) * (1 - 1e-6))
    if wl.category_count > 1:
        r3, w3 = GammaSiteRateModel(alpha=0.5 * (1 - 1e-6), gamma_categories=wl.category_count).category_rates_and_proportions()
    else:
        r3, w3 = wl.cat_rates, wl.cat_weights
    out.append((eig3, wl.freqs, np.asarray(r3, dtype=float), np.asarray(w3, dtype=float)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="A", choices=["A", "B", "C", "D", "E"])
    ap.add_argument("--caller", default="tdl", choices=["tdl", "btl"],
                    help="tdl: TreeDataLikelihood protocol (default); btl: BeagleTreeLikelihood also reads the site log-likelihoods back every evaluation")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink taxa/patterns (development only; 1.0 = the metric's config)")
    ap.add_argument("--tree", default="coalescent", choices=["coalescent", "yule", "caterpillar"])
    ap.add_argument("--real", default="", choices=["", "benchmark1", "benchmark2"],
                    help="config D on a REAL alignment of the reference's examples/Benchmarks (tests/golden/<name>_patterns.npz) instead of the synthetic stand-in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--patterns", type=int, default=0, help="development: keep only the first N patterns (size of one shard of an N-GPU job)")
    ap.add_argument("--force-sharded", action="store_true", help="development: take the multi-GPU code path (process group, device-side sum, all-reduce) even with one rank")
    ap.add_argument("--cache", default="/tmp/beagle_mi355_cache", help="directory for the generated workload ('' = off)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="patterns in the CPU-baseline sample (0 = sized for ~10-20 s of CPU work)")
    ap.add_argument("--rescaling", default="dynamic", choices=["dynamic", "always", "none"],
                    help="dynamic (default, the metric's protocol): steady state = read mode, every 100th evaluation recomputes the factors; "
                         "always: PartialsRescalingScheme ALWAYS, every evaluation rescales in write mode")
    ap.add_argument("--route", default="ranks", choices=["ranks", "library"],
                    help="ranks: one process per GPU + torch.distributed all-reduce (default); library: one process, the engine's resource G+1")
    ap.add_argument("--no-library-route", action="store_true", help="do not append the in-library route's run to the line")
    ap.add_argument("--spinup", type=float, default=-1.0,
                    help="seconds of untimed evaluations BEFORE the --warmup steps (development: how long the device takes to reach its "
                         "steady clocks after the idle seconds of workload generation); default: none")
    ap.add_argument("--step-times", action="store_true", help="development: the wall-clock time of every timed step in the line (step_ms)")
    ap.add_argument("--no-side-records", action="store_true", help="no shard_point / partial_update sub-records (the rocprofv3 re-runs pass this)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not re-run under rocprofv3 for roofline.traffic (the re-runs themselves pass this)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="time limit of the CPU baseline's multi-thread loop (the one-core loop gets 0.8 of it)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="do not append the other BASELINE configs (B, C, D on the reference's benchmark1 alignment, E) to the default line")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU check of the N-rank bring-up only (gloo, no engine, no GPU): tests/test_host_and_abi.py")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.route == "ranks" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))          # started plainly: bring up the N ranks ourselves
    if args.selftest_launcher:
        return selftest_launcher(args)

    # stdout carries ONE line, the JSON line: everything else that writes to file descriptor 1 in this process — RCCL's version
    # banner at communicator creation is C stdio — goes to stderr from here on, and the line is written to the real stdout last
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import beast_mcmc_amd as bm
    from beast_mcmc_amd.sharding import ShardedTreeLikelihood
    from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_DYNAMIC

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.route == "library":
        if world > 1:
            raise SystemExit("--route library is ONE process driving all GPUs; do not start it under torch.distributed.run")
    elif world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to report a line whose n_gpus is not what was asked for" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    if torch.cuda.device_count() < args.gpus:
        raise SystemExit("--gpus %d but only %d GPU(s) are visible" % (args.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:          # --force-sharded without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29555", RANK="0", WORLD_SIZE="1")
        dist.init_process_group(backend="nccl", device_id=device)

    import gc
    gc.collect()
    gc.disable()                       # (timed_loop; the process exits right after the line is printed)
    t_gen = time.time()
    makers = {"A": lambda: bm.synth.config_a(scale=args.scale, tree_kind=args.tree),
              "B": lambda: bm.synth.config_b(scale=args.scale),
              "C": lambda: bm.synth.config_c(scale=args.scale),
              "D": lambda: bm.synth.config_d(categories=1),
              "E": lambda: bm.synth.config_e(scale=args.scale)}
    if args.cache:
        os.makedirs(args.cache, exist_ok=True)
    cache = os.path.join(args.cache, "wl_%s_%g_%s.pkl" % (args.config, args.scale, args.tree)) if args.cache else None
    if cache and world > 1 and rank != 0:
        dist.barrier()                                  # rank 0 generates, the others read its file
    if args.real:
        args.config, cache = "D", None
        makers["D"] = lambda: bm.synth.from_pattern_fixture(os.path.join(ROOT, "tests", "golden", args.real + "_patterns.npz"))
    wl = bm.synth.cached(cache, makers[args.config])
    if cache and world > 1 and rank == 0:
        dist.barrier()
    t_gen = time.time() - t_gen
    if args.patterns and args.config != "E":
        wl = wl.shard(0, min(args.patterns, wl.pattern_count))
    res = (local_rank + 1,)                              # resource numbering: 0 = CPU (absent), 1..G = GPUs as THIS process sees them
    if args.route == "library":
        # resource G+1 = "all GPUs, pattern-sharded" (csrc/sharded.cpp); BEAGLE_MI355_SHARDS = N keeps it to the first N devices
        os.environ["BEAGLE_MI355_SHARDS"] = str(args.gpus)
        res = (torch.cuda.device_count() + 1,)

    if args.config == "E":
        out = bench_partitioned(args, bm, wl, rank, world, dist, device, res, t_gen)
    else:
        out = bench_single(args, bm, wl, rank, world, dist, device, res, t_gen, sharded,
                           ShardedTreeLikelihood, BeagleTreeLikelihood, RESCALE_DYNAMIC)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and out is not None and args.route == "ranks" and not args.no_library_route and args.config != "E":
        # the other multi-GPU route on the same workload (a shorter run), after this route's instances and process group are gone
        try:
            if world > 1:
                # N GPUs driven from ONE process (N host threads, ncclCommInitAll): in a process of its own with a time limit, so
                # that nothing it does — including not returning — can cost the line of the route that was asked for
                out["library_route"] = library_route_subprocess(args, world)
            else:
                out["library_route"] = library_route(args, bm, wl, world, torch, device, BeagleTreeLikelihood, RESCALE_DYNAMIC)
        except Exception as e:                                        # noqa: BLE001  (must not cost the main line)
            out["library_route"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if (rank == 0 and out is not None and world == 1 and args.route == "ranks" and args.config == "A" and args.scale == 1.0
            and not args.patterns and not args.no_side_records):
        try:
            out["shard_point"] = shard_point(args, bm, wl, torch, device, res, ShardedTreeLikelihood, BeagleTreeLikelihood, RESCALE_DYNAMIC)
        except Exception as e:                                        # noqa: BLE001
            out["shard_point"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if (rank == 0 and out is not None and world == 1 and args.route == "ranks" and args.config == "A" and args.scale == 1.0
            and not args.patterns and not args.no_side_records and not args.no_other_configs):
        try:
            out["other_configs"] = other_configs(args)
        except Exception as e:                                        # noqa: BLE001
            out["other_configs"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # ONE JSON line, and it is the LAST thing on stdout: libraries that print through C stdio (RCCL's version banner on the
    # multi-GPU path) are flushed first, so nothing of theirs can follow the line when the process exits
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:                       # noqa: BLE001
        pass
    sys.stdout.flush()
    if rank == 0 and out is not None:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    return out


def other_configs(args, budget_s=420.0):
    """BASELINE.json's other configurations on the same box, in the same run as the headline: B (20 states), C (61 states), D on
    the reference's own examples/Benchmarks/benchmark1.xml alignment (tests/golden fixture) and E (partitioned), each as a child
    `bench.py --config X` of >= 50 timed steps with its own kernel timer, in-run rocprofv3 traffic passes and CPU baseline (a
    shorter one: --cpu-seconds 4), cut down to the figures a reader compares.  A child that fails or overruns costs only its own
    entry; the whole record stops starting children once `budget_s` is spent."""
    import subprocess
    runs = [("B", [], 50, 5), ("C", [], 50, 5), ("D", ["--real", "benchmark1"], 300, 20), ("E", [], 300, 20)]      # (config, arguments, steps, warm-up steps)
    rec, t_start = {}, time.time()
    for name, extra, steps, warm in runs:
        left = budget_s - (time.time() - t_start)
        if left < 30.0:
            rec[name] = {"error": "not started: the record's time budget (%.0f s) was spent" % budget_s}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(steps), "--warmup", str(warm), "--no-library-route",
               "--no-side-records", "--no-other-configs", "--cpu-seconds", "4", "--cache", args.cache] + extra
        t0 = time.time()
        try:
            cp = subprocess.run(cmd, capture_output=True, text=True, timeout=min(left, 180.0))
            lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if cp.returncode != 0 or not lines:
                rec[name] = {"error": "rc %d: %s" % (cp.returncode, (cp.stderr or "")[-300:])}
                continue
            d = json.loads(lines[-1])
        except subprocess.TimeoutExpired:
            rec[name] = {"error": "timed out"}
            continue
        rf, cb = d.get("roofline") or {}, d.get("cpu_baseline") or {}
        rec[name] = {
            "workload": (d.get("config") or {}).get("workload"), "data": d.get("data"),
            "value": d.get("value"), "unit": d.get("unit"), "steps": d.get("steps"), "ms_per_step": d.get("ms_per_step"),
            "ms_per_step_median": d.get("ms_per_step_median"), "lnL": d.get("lnL"),
            "roofline": {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                "bytes_per_eval", "kernel_us_per_eval", "launches_per_eval", "fp64_TFLOPs")},
            "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "gpu_vs_cpu_site_lnL_max_rel_err", "gpu_vs_cpu_partition_lnL_max_rel_err") if k in cb},
            "kernel_source_hash": d.get("kernel_source_hash"), "wall_s": round(time.time() - t0, 1),
        }
    rec["what"] = ("child runs of this bench.py in the same invocation (B, C: --steps 50 --warmup 5; D, E — 0.1 ms per step —: --steps 300 --warmup 20; CPU baseline limited to 4 s); "
                   "D = the reference's benchmark1 alignment, 593 unique patterns")
    return rec


def launch_ranks(n):
    """`python bench.py --gpus N` started plainly: run the same command line under torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1 (the container hostname may not resolve).  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def selftest_launcher(args):
    """The rank bring-up without the engine (no GPU in the build container): every rank joins a gloo group, one all-reduce,
    rank 0 prints a line whose n_gpus is the world size it really ran with."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to report a line whose n_gpus is not what was asked for" % (args.gpus, world))
    total = float(rank + 1)
    if world > 1:
        dist.init_process_group(backend="gloo")
        t = torch.tensor([total], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    out = {"metric": "launcher selftest", "n_gpus": world, "rank_sum": total}
    if rank == 0:
        print(json.dumps(out), flush=True)
    return out


def timed_loop(torch, device, dist, steps, step, after_barrier=None):
    """Barrier + device synchronisation on both sides, MAX over ranks.  The Python collector is off while bench.py measures
    (main() collects once and disables it: a generation-2 collection of a process with torch loaded takes ~40 ms — ten
    evaluations' worth, triggered by the harness' own ctypes argument objects, and an idle gap in which the device drops its
    clocks: the first ~30 evaluations after such a gap run up to 20 % slower, profiles/r04_experiments.txt).
    after_barrier: called between the opening barrier and the clock (cheap bookkeeping only)."""

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    barrier()
    if after_barrier is not None:
        after_barrier()
    t0 = time.perf_counter()
    v = None
    for i in range(steps):
        v = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed, v


def moved_bytes(stats, p, c, s):
    """HBM bytes the 4-state design HAS to move for what the engine actually enqueued (its own counters): every stored
    node is written once, every child it could not keep in registers is read once, plus tip-state, scale-factor and
    reciprocal vectors.  (Algorithmic bytes count every node as written and every internal child as read.)"""
    b = p * s * c * 8
    return (stats["stored"] + stats["mem_reads"]) * b + stats["tip_reads"] * p + stats["scale_reads"] * 8 * p + stats["scale_writes"] * 16 * p


def profiled_traffic(config):
    """roofline.traffic: HBM bytes per evaluation from the rocprofv3 PMC passes of profiles/collect.sh (FETCH_SIZE x 2 on
    gfx950 + WRITE_SIZE), accepted only when the profile was taken on THIS build of the kernels."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None, "no profile under profiles/"
    try:
        rec = json.load(open(path)).get(config)
    except Exception as e:                         # noqa: BLE001
        return None, "unreadable profile: %s" % e
    if not rec:
        return None, "no profile for config %s" % config
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None, "profiles/hbm_traffic.json is from another build (%s, now %s): re-run profiles/collect.sh" % (
            rec.get("kernel_source_hash"), kernel_source_hash())
    return rec, "profiles/hbm_traffic.json (%s)" % rec.get("source", "")


def live_traffic(args, kernel):
    """roofline.traffic measured IN THIS RUN: two short re-runs of this very command line under rocprofv3, one per counter
    (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are never combined with a trace — MI355X_MICROARCH.md), summed over
    every dispatch of the hot kernel and divided by the evaluations the re-run executed.  FETCH_SIZE is doubled (gfx950
    tallies the 128-byte requests of a streaming read at 64 bytes), both are in units of 1024 bytes (profiles/summarize.py).
    -> (record, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled"
    base = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-library-route",
            "--no-live-traffic", "--no-side-records", "--config", args.config, "--caller", args.caller, "--scale", str(args.scale), "--tree", args.tree,
            "--rescaling", args.rescaling, "--cache", args.cache]
    if args.patterns:
        base += ["--patterns", str(args.patterns)]
    if args.real:
        base += ["--real", args.real]
    rec = {"kernel": kernel}
    t0 = time.time()
    for counter, factor, key in (("FETCH_SIZE", 2.0, "read"), ("WRITE_SIZE", 1.0, "write")):
        tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "pmc", "--"] + base,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            evals = max(1, int(json.loads(lines[-1]).get("evaluations_total", 1)))
            total, n = 0.0, 0
            for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        total += float(row["Counter_Value"]); n += 1
            if n == 0:
                return None, "no dispatch of %s in the %s pass" % (kernel, counter)
            rec["%s_bytes_per_eval" % key] = int(total * 1024.0 * factor / evals)
            rec["%s_dispatches" % key] = n
        except subprocess.TimeoutExpired:
            return None, "rocprofv3 --pmc %s timed out" % counter
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    rec["bytes_per_eval"] = rec["read_bytes_per_eval"] + rec["write_bytes_per_eval"]
    return rec, "measured in this run: rocprofv3 --pmc FETCH_SIZE (x2) / --pmc WRITE_SIZE, separate passes of 4 evaluations each, %.0f s" % (time.time() - t0)


def traffic_for(args, kernel, world, rank):
    """roofline.traffic: live (live_traffic) on a 1-GPU run when rocprofv3 is there, else replayed from the committed profile
    of the same build (profiled_traffic) and labelled so."""
    if world == 1 and rank == 0 and not args.no_live_traffic:
        rec, note = live_traffic(args, kernel)
        if rec:
            return rec, note
        live_note = note
    else:
        live_note = "not measured live (%s)" % ("child of a profiled run or --no-live-traffic" if args.no_live_traffic else "multi-GPU run")
    if args.patterns or args.scale != 1.0 or args.real or args.rescaling != "dynamic":
        return None, live_note + "; no replay: not the profiled size / protocol"
    rec, note = profiled_traffic(args.config)
    return rec, ("REPLAYED from " + note if rec else note) + " [" + live_note + "]"


def bench_single(args, bm, wl, rank, world, dist, device, res, t_gen, sharded, ShardedTreeLikelihood, BeagleTreeLikelihood, RESCALE_DYNAMIC):
    import numpy as np
    import torch
    # DYNAMIC rescaling with beagle.delay.scaling off: scalers are recomputed on the first evaluation and every
    # `beagle.rescale` = 100 evaluations, every other evaluation READS the stored factors.
    # (With the delay on, this realistic low-divergence tree never underflows in fp64 and scaling would never
    # switch on: fewer bytes, an easier benchmark.)
    from beast_mcmc_amd.treelikelihood import RESCALE_ALWAYS
    from beast_mcmc_amd.treelikelihood import RESCALE_NONE
    kw = dict(resource_list=res, rescaling={"always": RESCALE_ALWAYS, "none": RESCALE_NONE}.get(args.rescaling, RESCALE_DYNAMIC), delay_rescaling=False)
    if sharded:
        tl = ShardedTreeLikelihood(wl, rank, world, dist=dist, device=device, **kw)
        local = tl.local
    else:
        tl = BeagleTreeLikelihood(wl, **kw)
        local = tl
    models = perturbed_models(bm, wl, args.config)
    handles = [local.model_handle(*m) for m in models]       # the parameter blocks as C pointers, made once (treelikelihood.py)
    site_buf = {"n": 0}

    def step(i):
        # one MCMC iteration's worth of host protocol (MarkovChain.java:207-263): storeState (buffer indices flip on the
        # next write), then a substitution-model + site-model move: everything dirty, a NEW eigen system and NEW rates
        # uploaded (the next of three models: perturbed_models), all 2T-2 matrices and all T-1 partials recomputed into the
        # alternate buffers
        local.storeState()
        local.apply_model(handles[i % 3])
        v = tl.getLogLikelihood()
        if args.caller == "btl":
            site_buf["n"] += local.getSiteLogLikelihoods().shape[0]
        return v

    # reach DYNAMIC steady state (first evaluation underflows and recomputes the scalers)
    lnl0 = step(0)
    step(1)
    spin_evals = 0
    if args.spinup > 0:
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.spinup:
            step(spin_evals)
            spin_evals += 1
    raw = bm.beagle.Beagle.__new__(bm.beagle.Beagle)
    raw.lib, raw._f, raw.instance = local.engine, local.engine.fn, local.instance
    # HIP events around the pruning launches of every TIMER_EVERY-th evaluation of the timed region (an event pair is two
    # barrier packets on the stream: 12 us of GPU idle per bracketed evaluation, profiles/r04_experiments.txt).  Armed HERE —
    # it synchronises and creates its event pairs — and restarted (free) when the timed region begins.
    raw.kernelTimer(TIMER_EVERY)

    # the other caller protocol, for the record (a shorter run of the same loop) — BEFORE the main measurement: whatever
    # bench.py measures first finds a device that has idled through the seconds of workload generation, and the main line
    # should not be the one that pays for that (step_ms of a cold start: profiles/r04_experiments.txt)
    other = None
    if world == 1:
        keep = args.caller
        args.caller = "btl" if keep == "tdl" else "tdl"
        for i in range(3):
            step(i)                                   # (its first calls allocate: read-back buffers, numpy arrays)
        n2 = max(30, args.steps // 4)
        e2, _ = timed_loop(torch, device, dist, n2, step)
        other = {"caller": args.caller, "evals_per_s": round(n2 / e2, 3), "ms_per_step": round(1e3 * e2 / n2, 4), "steps": n2}
        args.caller = keep

    for i in range(args.warmup):
        step(i)
    step_ms = []
    if args.step_times:
        inner = step

        def step(i, inner=inner):                    # noqa: F811
            t = time.perf_counter()
            v = inner(i)
            step_ms.append(round(1e3 * (time.perf_counter() - t), 4))
            return v

    def restart():
        raw.kernelTimerRestart()
        raw.kernelTimerCalls()

    # (the wall-clock time of every timed step: the median beside the mean says whether the mean is one slow call's — two
    # perf_counter reads per step, ~0.1 us)
    per_step = []
    plain_step = step

    def step(i, inner=plain_step):                   # noqa: F811
        t = time.perf_counter()
        v = inner(i)
        per_step.append(time.perf_counter() - t)
        return v

    elapsed, lnl = timed_loop(torch, device, dist, args.steps, step, after_barrier=restart)
    per_step.sort()
    stats = raw.walkStats()
    stats["fused_cherries"] = raw.walkLaunchInfo()["fused_cherries"]      # (micro-operations evaluated inside their consumers' stages: not stages of their own)
    kernel_ms, launches = raw.kernelTimer(False)
    timed_calls = None
    rccl = None
    if dist is not None:
        # every rank's own kernel time and what RCCL itself says about the communicator the evaluations went through
        timed_here = max(1, raw.kernelTimerCalls())
        mine = torch.tensor([kernel_ms * 1e3 / timed_here, float(raw.commRanks()), 1.0 if getattr(tl, "collective", "") == "engine" else 0.0],
                            dtype=torch.float64, device=device)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        rows = [g.cpu().tolist() for g in gathered]
        engine_route = all(r[2] == 1.0 for r in rows)
        rccl = {"ranks": int(min(r[1] for r in rows)) if engine_route else int(dist.get_world_size()),
                "route": "engine" if engine_route else "torch-fallback",
                "ranks_source": "ncclCommCount of every rank's engine communicator (minimum over ranks)" if engine_route
                                else "torch.distributed world size (the engine-side communicator was not used)",
                "kernel_us_per_eval_by_rank": [round(r[0], 2) for r in rows]}
        timed_calls_cached = timed_here
        kms = torch.tensor([kernel_ms], dtype=torch.float64, device=device)
        dist.all_reduce(kms, op=dist.ReduceOp.MAX)
        kernel_ms = float(kms.item())
    if timed_calls is None:
        timed_calls = timed_calls_cached if dist is not None else max(1, raw.kernelTimerCalls())
    evals_per_s = args.steps / elapsed
    counters = local.counters()

    out = None
    if rank == 0:
        n_gpus = args.gpus if args.route == "library" else world
        if args.route == "library":                       # the engine's counters are shard 0's, the kernel time the slowest shard's
            from beast_mcmc_amd.inputs import patterns as _pat
            shard = wl.shard(*_pat.shard_bounds(wl.pattern_count, n_gpus)[0])
        else:
            shard = wl.shard(*tl.range) if sharded else wl
        s_, p_, c_ = shard.state_count, shard.pattern_count, shard.category_count
        alg = prune_bytes_per_eval(shard)                     # SURVEY 8d algorithmic bytes of this rank's pruning
        kernel_s = kernel_ms * 1e-3 / timed_calls
        walk = s_ == 4 and stats["walks"] > 0
        if walk:
            moved = moved_bytes(stats, p_, c_, s_) / max(1, args.steps)
            kname = "k_walk4_fast" if stats.get("fast_walks", 0) * 2 > stats["walks"] else "k_walk4"
            launches_per_eval = stats["walks"] / max(1, args.steps)
        else:
            # the level kernels store and re-read every node they compute (= the algorithmic bytes), minus the tip-tip nodes
            # a <= 20-state instance defines instead of storing (engine counters)
            moved = moved_bytes(stats, p_, c_, s_) / max(1, args.steps) if stats["micro_ops"] > 0 else alg
            kname = (("k_walkT32" if stats["walks"] > 0 else "k_pruneTiled<5>") if 16 <= s_ <= 20 else
                     ("k_walkT64" if stats["walks"] > 0 else "k_pruneTiled<16>") if s_ <= 64 else "k_pruneGeneral")
            launches_per_eval = launches / timed_calls
        achieved = moved / kernel_s / 1e9 if kernel_s > 0 else 0.0
        prof, prof_note = traffic_for(args, "k_walk" if stats["walks"] > 0 else "k_prune", world, rank)
        roofline = {
            "bound": "hbm", "kernel": kname,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": int(prof["bytes_per_eval"]) if prof else None, "traffic_source": prof_note,
            "traffic_frac_of_peak": round(prof["bytes_per_eval"] / kernel_s / 1e9 / HBM_PEAK_GBS, 4) if prof and kernel_s > 0 else None,
            "bytes_per_eval": int(moved), "bytes_basis": "engine counters: stored + re-read partials, tip, scale vectors" if stats["micro_ops"] > 0 else "algorithmic (every node stored and re-read)",
            # context for `frac`: what a plain grid-strided copy kernel sustains on this chip (tools/hbm_write_probe.hip,
            # profiles/r02_probes.txt: 2 x 2.6 TB/s; the guide's best float4 copy: 6.3 TB/s)
            "copy_rate_GBs_probe": 5200.0,           # NOT measured in this run: tools/hbm_write_probe.hip, profiles/r02_probes.txt
            "algorithmic_bytes_per_eval": int(alg), "effective_GBs": round(alg / kernel_s / 1e9, 1) if kernel_s > 0 else None,
            "kernel_us_per_eval": round(kernel_s * 1e6, 2), "launches_per_eval": round(launches_per_eval, 2),
            "kernel_timed_evaluations": int(timed_calls),
            "kernel_time_fraction_of_step": round(kernel_s * evals_per_s, 4),
        }
        if stats["micro_ops"] > 0:
            roofline["per_eval"] = {k: round(v / max(1, args.steps), 1) for k, v in stats.items()}
        # arithmetic side (matters for 61 states): 2*S*S flops per internal child per (pattern, category) + S products;
        # fp64 matrix/vector peak 78.6 TFLOP/s nominal, 73.9 measured for mfma_f64_4x4x4 (profiles/)
        tr = shard.tree
        n_int_children = sum(1 for n in range(shard.tip_count, 2 * shard.tip_count - 1)
                             for ch in (int(tr.left[n]), int(tr.right[n])) if ch >= shard.tip_count)
        flops = (n_int_children * 2.0 * s_ * s_ + (shard.tip_count - 1) * s_) * p_ * c_
        tflops = flops / kernel_s / 1e12 if kernel_s > 0 else 0.0
        roofline["fp64_TFLOPs"] = round(tflops, 2)
        roofline["fp64_frac_of_78.6"] = round(tflops / 78.6, 4)
        if tflops / 78.6 > achieved / HBM_PEAK_GBS:            # the compute roof is the nearer one (codon models)
            # (which pipe: the 4-state walk is DPP v_fmac_f64 on the vector ALU — no matrix-core instruction in it; 16..64 states
            # run v_mfma_f64_4x4x4.  Both pipes have the same fp64 peak on this chip.)
            roofline.update({"bound": "fp64-valu" if s_ == 4 else "mfma", "achieved": round(tflops, 2), "peak": 78.6, "unit": "TFLOP/s",
                             "frac": round(tflops / 78.6, 4), "hbm_GBs": round(achieved, 1)})
        collective_name = ("ncclAllReduce inside the engine, on its stream (RCCL over xGMI; torch.distributed only carried the communicator id)"
                           if sharded and getattr(tl, "collective", "") == "engine" else "torch.distributed over RCCL")
        cpu = None
        if n_gpus == 1 and args.route == "ranks" and not args.no_cpu_baseline:
            cpu = cpu_baseline(bm, wl, args.cpu_sample, tl, seconds=args.cpu_seconds)
        partial = None           # (after the CPU cross-check, which reads this instance's site values of the unmoved tree)
        if world == 1 and args.route == "ranks" and not sharded and args.config in ("A", "D") and not args.no_side_records:
            try:
                partial = partial_update_point(bm, wl, tl, raw, handles=handles)
            except Exception as e:                                        # noqa: BLE001  (must not cost the main line)
                partial = {"error": "%s: %s" % (type(e).__name__, e)}
        out = {
            "metric": "full-tree lnL evals/sec (GTR+G4, 1e5 patterns)" if args.config == "A" else "full-tree lnL evals/sec",
            "value": round(evals_per_s, 3), "unit": "evals/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "ms_per_step_median": round(1e3 * per_step[len(per_step) // 2], 4) if per_step else None,
            "ms_per_step_max": round(1e3 * per_step[-1], 4) if per_step else None,
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "rccl": rccl,
            "data": ("real alignment of examples/Benchmarks/%s.xml (tests/golden), seeded coalescent tree" % args.real) if args.real else "synthetic",
            "route": args.route,
            "config": {"workload": "%s: %d taxa x %d unique patterns, %d states, %d rate categories, %s tree (%d dependency levels), "
                                   "%s, new eigen system + rates every step"
                                   % (wl.name, wl.tip_count, wl.pattern_count, wl.state_count, wl.category_count, args.tree, wl.tree.depth(),
                                      {"always": "ALWAYS rescaling (write mode every evaluation)", "none": "NO rescaling (development)"}.get(args.rescaling, "DYNAMIC rescaling steady state")),
                       "caller": args.caller, "patterns_per_gpu": p_, "parallelism": ("pattern-shard x%d + 1 all-reduce (%s), one process per GPU" % (n_gpus, collective_name)) if args.route == "ranks"
                                      else "pattern-shard x%d inside the library (resource G+1, ncclAllReduce), one process" % n_gpus,
                       "ops_per_eval": int(counters["last_op_count"]), "matrices_per_eval": int(counters["last_branch_count"])},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "other_caller": other,
            "partial_update": partial,
            "spinup_evaluations": spin_evals, "step_ms": step_ms[:args.steps] if args.step_times else None,
            "lnL": lnl, "lnL_first_eval": lnl0, "hbm_bytes_resident": int(raw.deviceBytes()),
            "evaluations_total": int(local.counters()["evaluations"]),
            "kernel_source_hash": kernel_source_hash(), "workload_generation_s": round(t_gen, 1),
        }
    tl.close()
    return out


def shard_point(args, bm, wl, torch, device, res, ShardedTreeLikelihood, BeagleTreeLikelihood, RESCALE_DYNAMIC, patterns=12500):
    """What ONE GPU of an 8-GPU job does per evaluation, measured on this GPU: the first `patterns` (= 1e5 / 8) patterns of the
    metric's alignment through the MULTI-GPU code path — sharding.py's ShardedTreeLikelihood, the engine's device-side sum, the
    RCCL all-reduce (a communicator of one rank: ncclAllReduce is issued and waited for like any other), the decision on the
    global value — the same step protocol as the main line.  It bounds the 2/4/8-GPU points of the metric from below: an 8-GPU
    evaluation cannot take less than this plus what the collective costs across real links."""
    shard = wl.shard(0, min(patterns, wl.pattern_count))
    kw = dict(resource_list=res, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    tl = ShardedTreeLikelihood(shard, 0, 1, dist=None, device=device, collective="engine", **kw)
    local = tl.local
    handles = [local.model_handle(*m) for m in perturbed_models(bm, shard, args.config)]

    def step(i):
        local.storeState()
        local.apply_model(handles[i % 3])
        return tl.getLogLikelihood()

    step(0); step(1)
    for i in range(10):
        step(i)
    n = max(200, args.steps)
    per = []

    def clocked(i):
        t = time.perf_counter()
        v = step(i)
        per.append(time.perf_counter() - t)
        return v

    # Two windows of n steps.  The first holds the chain's first rescaling cycles (DYNAMIC recomputes the factors every 100
    # evaluations, flipping between two scale-buffer sets: six operation lists in all, each planned, resolved and uploaded once —
    # ~2 ms of host work that a chain of millions of steps pays twice); the second is the steady state, rescaling evaluations
    # included (two per 200 steps), every program out of the plan cache.  `ms_per_step` is the second one.
    cold, _ = timed_loop(torch, device, None, n, clocked)
    del per[:]
    elapsed, lnl = timed_loop(torch, device, None, n, clocked)
    stats_eval = local.counters()["evaluations"]
    tl.close()
    # the same steps on a plain single-GPU instance of the same shard: the value must be the same double
    ref = BeagleTreeLikelihood(shard, **kw)
    rh = [ref.model_handle(*m) for m in perturbed_models(bm, shard, args.config)]
    v = None
    for i in [0, 1] + list(range(10)) + list(range(n)) + list(range(n)):
        ref.storeState(); ref.apply_model(rh[i % 3]); v = ref.getLogLikelihood()
    ref.close()
    per.sort()
    return {"what": "one GPU's share of an 8-GPU evaluation (pattern shard of the metric's alignment), multi-GPU code path",
            "patterns": shard.pattern_count, "value": round(n / elapsed, 3), "unit": "evals/s", "steps": n,
            "ms_per_step": round(1e3 * elapsed / n, 4), "ms_per_step_median": round(1e3 * per[len(per) // 2], 4),
            "ms_per_step_max": round(1e3 * per[-1], 4),
            "ms_per_step_first_window": round(1e3 * cold / n, 4),
            "windows": "two of %d steps behind 12 untimed ones; the first holds the chain's first two rescaling cycles (their six operation lists are planned and uploaded once), ms_per_step is the second: steady state, its two rescaling evaluations included" % n,
            "collective": "ncclAllReduce(1 double) inside the engine on its stream, communicator of 1 rank, inside every timed step",
            "lnL": lnl, "lnL_equals_unsharded_instance": bool(lnl == v), "evaluations_total": int(stats_eval)}


def partial_update_point(bm, wl, tl, raw, moves=300, handles=None):
    """What a chain issues most (TreeDataLikelihood.java:247-260): ONE node height changes, the path from that node to the root
    is recomputed — its three branch matrices, then the operations up to the root —, half of the proposals are rejected
    (restoreState: index flips only).  Microseconds per move (proposal + the 50 % restore), operations and bytes moved per move."""
    import numpy as np
    rng = np.random.default_rng(5)
    t_, n_ = wl.tree.tip_count, wl.tree.node_count
    height = np.array(wl.tree.height, dtype=float)
    parent = np.full(n_, -1)
    for n in range(t_, n_):
        parent[int(wl.tree.left[n])] = n; parent[int(wl.tree.right[n])] = n

    def propose():
        node = int(rng.integers(t_, n_))
        while parent[node] < 0:
            node = int(rng.integers(t_, n_))
        lo = max(height[int(wl.tree.left[node])], height[int(wl.tree.right[node])])
        hi = height[parent[node]]
        return node, lo + (hi - lo) * float(rng.uniform(0.05, 0.95))          # stays between its children and its parent

    def run(k):
        for _ in range(k):
            node, h = propose()
            tl.storeState()
            tl.set_node_height(node, h)
            tl.getLogLikelihood()
            if rng.random() < 0.5:
                tl.restoreState()
                tl.restore_node_height(node, float(height[node]))      # (the tree model's own restore)
                tl.getLogLikelihood()                                   # known: no engine call
            else:
                height[node] = h

    run(30)
    raw.kernelTimer(False)
    c0 = tl.counters()
    t0 = time.perf_counter()
    run(moves)
    dt = time.perf_counter() - t0
    c1 = tl.counters()
    stats = raw.walkStats()
    p_, c_, s_ = wl.pattern_count, wl.category_count, wl.state_count
    # A chain mixes the two: after accepted moves the nodes of their paths sit in the OTHER buffer of their pair, so the next
    # model move's full-evaluation list names a combination of buffers no earlier list had — the engine plans, resolves and uploads
    # it from scratch (the main line's chain of full evaluations alternates between two lists and always finds its program
    # resident).  Three accepted moves, then a model move: milliseconds per such full evaluation, beside the main line's.
    mixed = None
    if handles:
        full = []
        for i in range(40):
            for _ in range(3):
                node, h = propose()
                tl.storeState(); tl.set_node_height(node, h); tl.getLogLikelihood(); height[node] = h
            tl.storeState()
            tl.apply_model(handles[i % 3])
            t1 = time.perf_counter()
            tl.getLogLikelihood()
            full.append(time.perf_counter() - t1)
        full.sort()
        mixed = {"what": "a model move (every node recomputed) behind three accepted node-height moves: an operation list the engine has not seen",
                 "ms_per_full_evaluation_median": round(1e3 * full[len(full) // 2], 4), "ms_per_full_evaluation_mean": round(1e3 * sum(full) / len(full), 4),
                 "evaluations": len(full)}
    # ... and the two mixed as a chain mixes them: nine node-height proposals (half of them rejected) to one model move, 400 proposals —
    # likelihood evaluations per second of such a chain, beside the headline's chain of model moves only
    chain_mixed = None
    if handles:
        n_prop, n_model, evals = 400, 0, 0
        t1 = time.perf_counter()
        for i in range(n_prop):
            if i % 10 == 9:
                tl.storeState()
                tl.apply_model(handles[n_model % 3]); n_model += 1
                tl.getLogLikelihood(); evals += 1
            else:
                node, h = propose()
                tl.storeState()
                tl.set_node_height(node, h)
                tl.getLogLikelihood(); evals += 1
                if rng.random() < 0.5:
                    tl.restoreState()
                    tl.restore_node_height(node, float(height[node]))
                    tl.getLogLikelihood()                               # (known: no engine call)
                else:
                    height[node] = h
        dtm = time.perf_counter() - t1
        chain_mixed = {"what": "a chain of 90 % node-height moves (half rejected) and 10 % model moves (every node recomputed, mostly on operation lists the engine has not seen)",
                       "proposals": n_prop, "model_moves": n_model, "evals_per_s": round(evals / dtm, 1), "us_per_proposal": round(1e6 * dtm / n_prop, 2)}
    return {"what": "one node-height move: path to the root recomputed, 50 % of the proposals restored",
            "full_evaluation_on_a_new_list": mixed, "chain_mixed": chain_mixed,
            "us_per_branch_move": round(1e6 * dt / moves, 2), "moves": moves,
            "ops_per_move": round((c1["operations"] - c0["operations"]) / moves, 2),
            "matrices_per_move": round((c1["matrix_updates"] - c0["matrix_updates"]) / moves, 2),
            "bytes_per_move": int(moved_bytes(stats, p_, c_, s_) / moves) if s_ == 4 else None,
            "micro_ops_per_move": round(stats["micro_ops"] / moves, 1), "stored_per_move": round(stats["stored"] / moves, 2)}


def library_route(args, bm, wl, n, torch, device, BeagleTreeLikelihood, RESCALE_DYNAMIC):
    """The same step through ONE instance of resource G+1 with n shards (csrc/sharded.cpp): what a JVM gets with
    -beagle_order <G+1> and no Java-side change.  One process; the library's host threads drive the GPUs."""
    if torch.cuda.device_count() < n:
        return {"error": "needs %d visible GPUs, this process sees %d" % (n, torch.cuda.device_count())}
    os.environ["BEAGLE_MI355_SHARDS"] = str(n)
    tl = BeagleTreeLikelihood(wl, resource_list=(torch.cuda.device_count() + 1,), rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    models = perturbed_models(bm, wl, args.config)
    handles = [tl.model_handle(*m) for m in models]

    def step(i):
        tl.storeState()
        tl.apply_model(handles[i % 3])
        return tl.getLogLikelihood()

    step(0); step(1)
    for i in range(max(10, min(args.warmup, 20))):
        step(i)
    # >= 50 timed steps whatever --steps says: a 10-step mean is at the mercy of ONE slow posted call (BENCH_r03: 673
    # evals/s here beside 1 302 on the rank route); every step is clocked, and the line carries the median and the
    # slowest step next to the mean
    n2 = max(50, args.steps // 2)
    per = []

    def clocked(i):
        t = time.perf_counter()
        v = step(i)
        per.append(time.perf_counter() - t)
        return v

    elapsed, lnl = timed_loop(torch, device, None, n2, clocked)
    tl.close()
    per.sort()
    return {"route": "library", "n_gpus": n, "value": round(n2 / elapsed, 3), "unit": "evals/s", "steps": n2,
            "ms_per_step": round(1e3 * elapsed / n2, 4), "ms_per_step_median": round(1e3 * per[len(per) // 2], 4),
            "ms_per_step_max": round(1e3 * per[-1], 4), "ms_per_step_p90": round(1e3 * per[(len(per) * 9) // 10], 4), "lnL": lnl}


def library_route_subprocess(args, n, limit=300):
    """library_route for n > 1 GPUs as `bench.py --route library --gpus n` in a child process (none of this rank's
    torch.distributed environment), killed after `limit` seconds."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                           "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.abspath(__file__), "--route", "library", "--gpus", str(n), "--steps", str(max(50, args.steps // 2)),
           "--warmup", str(min(args.warmup, 10)), "--config", args.config, "--scale", str(args.scale), "--tree", args.tree,
           "--rescaling", args.rescaling, "--cache", args.cache, "--no-cpu-baseline", "--no-live-traffic", "--no-library-route"]
    if args.patterns:
        cmd += ["--patterns", str(args.patterns)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit)
    except subprocess.TimeoutExpired:
        return {"error": "no result within %d s (child killed)" % limit}
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "child exited with %d: %s" % (r.returncode, (r.stderr or "").strip().splitlines()[-1:] or "")}
    d = json.loads(lines[-1])
    return {"route": "library", "n_gpus": d.get("n_gpus", n), "value": d["value"], "unit": d["unit"], "steps": d["steps"],
            "ms_per_step": d["ms_per_step"], "lnL": d.get("lnL"), "how": "child process"}


def bench_partitioned(args, bm, pw, rank, world, dist, device, res, t_gen):
    """Config E: four partitions on ONE instance per GPU (MultiPartitionDataLikelihoodDelegate's protocol); with N GPUs every
    partition's pattern range is cut into N contiguous blocks and the partitionCount per-partition values are all-reduced."""
    import numpy as np
    import torch
    from beast_mcmc_amd.multipartition import MultiPartitionTreeLikelihood
    local_pw = pw.shard(rank, world) if world > 1 else pw
    tl = MultiPartitionTreeLikelihood(local_pw, resource_list=res)
    k = len(pw.parts)
    buf = torch.zeros(k, dtype=torch.float64, device=device)
    rates0 = np.ones(pw.tree.node_count)

    def step(i):
        tl.set_branch_rates(rates0 * (1.0 + 1e-6 * (i & 1)))       # a clock-rate move: every matrix of every partition changes
        by_part, total = tl.calculate()
        if dist is not None:
            buf.copy_(torch.from_numpy(by_part))
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            total = float(buf.sum().item())
        return total

    lnl0 = step(0)
    tl.b.kernelTimer(TIMER_EVERY)                      # (armed before the warm-up, restarted for free when the timed region begins)
    for i in range(args.warmup):
        step(i)
    per_step = []

    def clocked(i, inner=step):                        # (two perf_counter reads per step: the median beside the mean)
        t = time.perf_counter()
        v = inner(i)
        per_step.append(time.perf_counter() - t)
        return v

    elapsed, lnl = timed_loop(torch, device, dist, args.steps, clocked, after_barrier=tl.b.kernelTimerRestart)
    per_step.sort()
    stats = tl.b.walkStats()
    stats["fused_cherries"] = tl.b.walkLaunchInfo()["fused_cherries"]
    kernel_ms, _ = tl.b.kernelTimer(False)
    timed_calls = max(1, tl.b.kernelTimerCalls())
    out = None
    if rank == 0:
        p_, c_ = local_pw.pattern_count, tl.C
        kernel_s = kernel_ms * 1e-3 / timed_calls
        # every partition's op touches only its own pattern range: bytes = sum over partitions
        moved = 0.0
        per_part = {key: v / max(1, args.steps) / k for key, v in stats.items()}
        for w in local_pw.parts:
            moved += moved_bytes(per_part, w.pattern_count, c_, 4)
        alg = sum(prune_bytes_per_eval(w) for w in local_pw.parts)
        achieved = moved / kernel_s / 1e9 if kernel_s > 0 else 0.0
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline_partitioned(bm, pw, seconds=args.cpu_seconds)
            # ... and its per-partition values beside this instance's for the same (unperturbed) branch rates
            tl.set_branch_rates(rates0)
            gpu_parts, _ = tl.calculate()
            cpu["gpu_vs_cpu_partition_lnL_max_rel_err"] = float(max(abs(a - b) / abs(b) for a, b in zip(gpu_parts, cpu.pop("partition_lnL"))))
        prof, prof_note = traffic_for(args, "k_walk4", world, rank)
        partial = None
        if world == 1 and not args.no_side_records:
            try:
                partial = partial_update_point_partitioned(tl, local_pw)
            except Exception as e:                                        # noqa: BLE001  (must not cost the main line)
                partial = {"error": "%s: %s" % (type(e).__name__, e)}
        out = {
            "metric": "full-tree lnL evals/sec", "value": round(args.steps / elapsed, 3), "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "ms_per_step_median": round(1e3 * per_step[len(per_step) // 2], 4) if per_step else None,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d taxa, unique patterns per partition %s, 4 states, %d rate categories, one instance, "
                                   "updatePartialsByPartition, new branch rates every step" % (pw.name, pw.tip_count, pw.pattern_counts, c_),
                       "patterns_per_gpu": p_, "parallelism": "pattern-shard x%d of every partition + 1 all-reduce of %d doubles" % (world, k)},
            "roofline": {"bound": "hbm", "kernel": "k_walk4", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": int(prof["bytes_per_eval"]) if prof else None, "traffic_source": prof_note,
                         "bytes_per_eval": int(moved), "algorithmic_bytes_per_eval": int(alg),
                         "kernel_us_per_eval": round(kernel_s * 1e6, 2), "kernel_time_fraction_of_step": round(kernel_s * args.steps / elapsed, 4),
                         "per_eval": {key: round(v / max(1, args.steps), 1) for key, v in stats.items()}},
            "cpu_baseline": cpu, "partial_update": partial, "lnL": lnl, "lnL_first_eval": lnl0, "kernel_source_hash": kernel_source_hash(),
            "evaluations_total": tl.evaluations, "workload_generation_s": round(t_gen, 1),
        }
    tl.close()
    return out


def partial_update_point_partitioned(tl, pw, moves=300):
    """partial_update_point for the partitioned instance (config E): one node height changes; per partition three branch matrices
    (one updateTransitionMatricesWithMultipleModels for all), the operations of the path to the root as 9-int tuples
    (updatePartialsByPartition), the root integration by partition; half of the proposals are taken back (offset flips only)."""
    import numpy as np
    rng = np.random.default_rng(5)
    tree = pw.tree
    t_, n_ = tree.tip_count, tree.node_count

    def run(k):
        for _ in range(k):
            node = int(rng.integers(t_, n_))
            while node == tree.root:
                node = int(rng.integers(t_, n_))
            lo = max(tree.height[int(tree.left[node])], tree.height[int(tree.right[node])])
            hi = tree.height[tree.parent[node]]
            tl.move_node_height(node, lo + (hi - lo) * float(rng.uniform(0.05, 0.95)))
            if rng.random() < 0.5:
                tl.restore_move()

    run(30)
    tl.b.kernelTimer(False)
    e0 = tl.evaluations
    tl.b.walkStats()
    t0 = time.perf_counter()
    run(moves)
    dt = time.perf_counter() - t0
    stats = tl.b.walkStats()
    k = len(pw.parts)
    return {"what": "one node-height move on the partitioned instance: path to the root recomputed for every partition, 50 % of the proposals restored",
            "us_per_branch_move": round(1e6 * dt / moves, 2), "moves": tl.evaluations - e0, "partitions": k,
            "micro_ops_per_move": round(stats["micro_ops"] / moves, 1), "stored_per_move": round(stats["stored"] / moves, 2),
            "matrices_per_move": 3 * k}


def _host_cpu_info():
    """What the box actually grants this process (the OpenMP default may exceed a container's CPU quota)."""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup:" + os.path.basename(path)] = open(path).read().strip()
            break
        except Exception:
            pass
    return info


def cpu_baseline(bm, wl, sample, gpu_tl, seconds=10.0):
    """The CPU oracle (oracle/beagle_cpu_oracle.c — a plain-C restatement, NOT beagle-lib) timed on this box's
    host cores on a bounded sample of the same workload: the full tree, `sample` of the P patterns (patterns
    are independent, so cost is linear in P).  Reported scaled to the full pattern count.  Also cross-checks the
    GPU's site log-likelihoods for the sampled patterns."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from beast_mcmc_amd.inputs.synth import Workload
    from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_DYNAMIC
    if not sample:                                            # ~10-20 s of CPU work whatever the state count
        per_pattern = wl.tip_count * wl.category_count * wl.state_count * wl.state_count
        sample = int(max(500, min(20000, 6.4e11 / per_pattern)))
    n = min(sample, wl.pattern_count)
    idx = np.sort(np.random.default_rng(0).choice(wl.pattern_count, size=n, replace=False))
    sub = Workload(wl.name + "-sample", wl.tree, wl.eig, wl.freqs, wl.cat_rates, wl.cat_weights,
                   np.ascontiguousarray(wl.tip_states[:, idx]), wl.weights[idx], wl.state_count)
    lib = helpers.oracle_library()
    # a container's CPU quota can be far below the core count OpenMP sees: helpers.oracle_library() caps the oracle's
    # thread count to helpers.granted_cpus(), so the baseline runs on what the box actually grants
    threads = lib.lib.oracle_threads()
    o = BeagleTreeLikelihood(sub, library=lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    o.getLogLikelihood()
    o.makeDirty(); o.getLogLikelihood()                       # steady state (read-mode scalers), as on the GPU
    reps, t0 = 0, time.perf_counter()
    while True:
        o.set_substitution_model(wl.eig, wl.freqs)
        o.getLogLikelihood()
        reps += 1
        dt = time.perf_counter() - t0
        if dt > seconds or reps >= 200:
            break
    site_cpu = o.getSiteLogLikelihoods()
    g = gpu_tl.local if hasattr(gpu_tl, "local") else gpu_tl
    g.storeState()
    g.set_substitution_model(wl.eig, wl.freqs)                # the unperturbed model, as the oracle ran
    g.set_site_model(wl.cat_rates, wl.cat_weights)
    gpu_tl.getLogLikelihood()
    site_gpu = g.getSiteLogLikelihoods()[idx]
    rel = float(np.max(np.abs(site_gpu - site_cpu) / np.abs(site_cpu)))
    o.close()
    sample_evals_per_s = reps / dt
    # the same port on ONE core (SURVEY 8d asks for both): a tenth of the sample, at most ~8 s
    n1 = max(1, n // 10)
    sub1 = Workload(wl.name + "-sample1", wl.tree, wl.eig, wl.freqs, wl.cat_rates, wl.cat_weights,
                    np.ascontiguousarray(wl.tip_states[:, idx[:n1]]), wl.weights[idx[:n1]], wl.state_count)
    lib.lib.oracle_set_threads(1)
    try:
        o1 = BeagleTreeLikelihood(sub1, library=lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
        o1.getLogLikelihood()
        reps1, t1 = 0, time.perf_counter()
        while True:
            o1.set_substitution_model(wl.eig, wl.freqs)
            o1.getLogLikelihood()
            reps1 += 1
            dt1 = time.perf_counter() - t1
            if dt1 > 0.8 * seconds or reps1 >= 50:
                break
        o1.close()
    finally:
        lib.lib.oracle_set_threads(int(threads))
    one_core = reps1 / dt1 * n1 / wl.pattern_count
    return {"value": round(sample_evals_per_s * n / wl.pattern_count, 4), "unit": "evals/s", "cores": int(threads),
            "kind": "port",
            "sample": "%d of %d patterns, full %d-taxon tree, %d evaluations in %.1f s on %d OpenMP threads; scaled by %d/%d"
                      % (n, wl.pattern_count, wl.tip_count, reps, dt, threads, n, wl.pattern_count),
            "host_cpus": _host_cpu_info(),
            "single_core_value": round(one_core, 5),
            "single_core_sample": "%d patterns, %d evaluations in %.1f s on 1 thread" % (n1, reps1, dt1),
            "gpu_vs_cpu_site_lnL_max_rel_err": rel}


def cpu_baseline_partitioned(bm, pw, seconds=10.0):
    """Config E on the host: the oracle evaluates the four partitions one after the other (whole alignment: it is small)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_NONE
    lib = helpers.oracle_library()
    threads = lib.lib.oracle_threads()
    tls = [BeagleTreeLikelihood(w, library=lib, rescaling=RESCALE_NONE, delay_rescaling=False) for w in pw.parts]
    part_lnl = [t.getLogLikelihood() for t in tls]
    reps, t0 = 0, time.perf_counter()
    while True:
        for t, w in zip(tls, pw.parts):
            t.set_substitution_model(w.eig, w.freqs)
            t.getLogLikelihood()
        reps += 1
        dt = time.perf_counter() - t0
        if dt > seconds or reps >= 200:
            break
    for t in tls:
        t.close()
    return {"partition_lnL": part_lnl, "value": round(reps / dt, 4), "unit": "evals/s", "cores": int(threads), "kind": "port",
            "sample": "the whole alignment (%d patterns in 4 partitions, %d taxa), %d evaluations in %.1f s on %d OpenMP threads"
                      % (pw.pattern_count, pw.tip_count, reps, dt, threads),
            "host_cpus": _host_cpu_info()}


if __name__ == "__main__":
    main()
