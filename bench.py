#!/usr/bin/env python3
"""Headline benchmark: full-tree log-likelihood evaluations per second, GTR+G4, 1000 taxa x 1e5 unique
patterns, fp64 (BASELINE.json `metric`, config "GTR+G4 nucleotide (4-state), 1000 taxa x 1e5 unique patterns").

One "step" = one full-tree evaluation driven through the C ABI exactly as BEAST drives it after a substitution-
model move: setEigenDecomposition + setCategoryRates + updateTransitionMatrices(2T-2 branches) +
updatePartials(T-1 level-ordered ops, steady-state DYNAMIC rescaling: read mode) + setCategoryWeights /
setStateFrequencies + calculateRootLogLikelihoods, with the scalar result read back on the host
(SURVEY 8d "Timing protocol").  All inputs are resident in HBM before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU): the 1e5 patterns are split into contiguous blocks
(Patterns.java:142-167), every rank evaluates its block, ONE RCCL all-reduce of the per-shard lnL per step.
Total work is fixed as N grows -> "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def prune_bytes_per_eval(wl):
    """Algorithmic HBM bytes of the pruning launches of ONE full-tree evaluation (SURVEY 8d / DESIGN.md §5):
    per op: destination write B + B per internal child (P bytes per compact tip child) + one 8-byte scale
    factor per pattern (read in steady-state DYNAMIC); B = P*S*C*8."""
    t, p = wl.tip_count, wl.pattern_count
    b = p * wl.state_count * wl.category_count * 8
    total = 0
    tree = wl.tree
    for n in range(t, 2 * t - 1):
        total += b + 8 * p
        for ch in (int(tree.left[n]), int(tree.right[n])):
            total += p if ch < t else b
    return total


def eval_bytes(wl):
    """Whole-evaluation algorithmic bytes: pruning + root read + weights, cumulative scalers, site-lnL write."""
    p = wl.pattern_count
    return prune_bytes_per_eval(wl) + p * wl.state_count * wl.category_count * 8 + 3 * p * 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="A", choices=["A", "B", "C", "D"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink taxa/patterns (development only; 1.0 = the metric's config)")
    ap.add_argument("--tree", default="coalescent", choices=["coalescent", "yule", "caterpillar"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--patterns", type=int, default=0, help="development: keep only the first N patterns (size of one shard of an N-GPU job)")
    ap.add_argument("--force-sharded", action="store_true", help="development: take the multi-GPU code path (process group, device-side sum, all-reduce) even with one rank")
    ap.add_argument("--cache", default="/tmp/beagle_mi355_cache", help="directory for the generated workload ('' = off)")
    ap.add_argument("--cpu-sample", type=int, default=20000, help="patterns in the CPU-baseline sample")
    args = ap.parse_args()

    import numpy as np
    import torch
    import beast_mcmc_amd as bm
    from beast_mcmc_amd.sharding import ShardedTreeLikelihood
    from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:          # --force-sharded without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29555", RANK="0", WORLD_SIZE="1")
        dist.init_process_group(backend="nccl", device_id=device)

    t_gen = time.time()
    makers = {"A": lambda: bm.synth.config_a(scale=args.scale, tree_kind=args.tree),
              "B": lambda: bm.synth.config_b(scale=args.scale),
              "C": lambda: bm.synth.config_c(scale=args.scale),
              "D": lambda: bm.synth.config_d(categories=1)}
    if args.cache:
        os.makedirs(args.cache, exist_ok=True)
    cache = os.path.join(args.cache, "wl_%s_%g_%s.pkl" % (args.config, args.scale, args.tree)) if args.cache else None
    if cache and world > 1 and rank != 0:
        dist.barrier()                                  # rank 0 generates, the others read its file
    wl = bm.synth.cached(cache, makers[args.config])
    if cache and world > 1 and rank == 0:
        dist.barrier()
    t_gen = time.time() - t_gen
    if args.patterns:
        wl = wl.shard(0, min(args.patterns, wl.pattern_count))

    # resource numbering: 0 = CPU (absent), 1..G = GPUs as THIS process sees them
    res = (local_rank + 1,)
    # DYNAMIC rescaling with beagle.delay.scaling off: scalers are recomputed on the first evaluation and every
    # `beagle.rescale` = 100 evaluations, every other evaluation READS the stored factors.
    # (With the delay on, this realistic low-divergence tree never underflows in fp64 and scaling would never
    # switch on: fewer bytes, an easier benchmark.)
    from beast_mcmc_amd.treelikelihood import RESCALE_DYNAMIC
    kw = dict(resource_list=res, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    if sharded:
        tl = ShardedTreeLikelihood(wl, rank, world, dist=dist, device=device, **kw)
        local = tl.local
    else:
        tl = BeagleTreeLikelihood(wl, **kw)
        local = tl

    def step():
        # one MCMC iteration's worth of host protocol (MarkovChain.java:207-263): storeState (buffer indices flip
        # on the next write), then a substitution-model + site-model move: everything dirty, eigen system and
        # rates re-uploaded, all 2T-2 matrices and all T-1 partials recomputed into the alternate buffers
        local.storeState()
        local.set_substitution_model(wl.eig, wl.freqs)
        local.set_site_model(wl.cat_rates, wl.cat_weights)
        return tl.getLogLikelihood()

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    # reach DYNAMIC steady state (first evaluation underflows and recomputes the scalers), then warm up
    lnl0 = step()
    step()
    for _ in range(args.warmup):
        step()
    raw = bm.beagle.Beagle.__new__(bm.beagle.Beagle)
    raw.lib, raw._f, raw.instance = local.engine, local.engine.fn, local.instance
    raw.kernelTimer(True)

    # The harness is Python: a generation-2 garbage collection of a process that has torch loaded takes ~40 ms
    # (measured: tools/shard_overhead.py per-step trace) — ten evaluations' worth — and is triggered by the ctypes
    # argument objects of the harness itself, not by anything on the measured path.  Collect now, pause the collector
    # for the timed region (no cycles are created in it), restore afterwards.
    import gc
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    lnl = 0.0
    for _ in range(args.steps):
        lnl = step()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()

    kernel_ms, launches = raw.kernelTimer(False)
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kms = torch.tensor([kernel_ms], dtype=torch.float64, device=device)
        dist.all_reduce(kms, op=dist.ReduceOp.MAX)
        kernel_ms = float(kms.item())

    evals_per_s = args.steps / elapsed
    counters = local.counters()

    out = None
    if rank == 0:
        shard = wl.shard(*tl.range) if sharded else wl
        pb = prune_bytes_per_eval(shard)                      # this rank's pruning bytes per evaluation
        launches_per_eval = launches / max(1, args.steps)
        kernel_s_per_eval = kernel_ms * 1e-3 / max(1, args.steps)
        achieved = pb / kernel_s_per_eval / 1e9 if kernel_s_per_eval > 0 else 0.0
        traffic = None
        tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get("bytes_per_launch")
            except Exception:
                traffic = None
        s_ = wl.state_count
        kname = "k_prune4<4,nt>" if s_ == 4 else ("k_pruneTiled<5>" if 16 <= s_ <= 20 else "k_pruneTiled<16>" if s_ <= 64 else "k_pruneGeneral")
        roofline = {
            "bound": "hbm", "kernel": kname,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic if args.config == "A" else None,
            "algorithmic_bytes_per_launch": round(pb / max(1.0, launches_per_eval)),
            "avg_launch_us": round(kernel_ms * 1e3 / max(1, launches), 2),
            "launches_per_eval": round(launches_per_eval, 2),
            "kernel_time_fraction_of_step": round(kernel_s_per_eval * evals_per_s, 4),
            "whole_eval_GBs": round(eval_bytes(shard) * evals_per_s / 1e9, 1),
        }
        # arithmetic side of the roofline (matters for 61 states): 2*S*S flops per internal child per (pattern, category)
        # + S products; fp64 matrix/vector peak 78.6 TFLOP/s (nominal), 73.9 measured for mfma_f64_4x4x4
        tr = shard.tree
        n_int_children = sum(1 for n in range(shard.tip_count, 2 * shard.tip_count - 1)
                             for ch in (int(tr.left[n]), int(tr.right[n])) if ch >= shard.tip_count)
        flops = (n_int_children * 2.0 * s_ * s_ + (shard.tip_count - 1) * s_) * shard.pattern_count * shard.category_count
        tflops = flops / kernel_s_per_eval / 1e12 if kernel_s_per_eval > 0 else 0.0
        roofline["fp64_TFLOPs"] = round(tflops, 2)
        roofline["fp64_frac_of_78.6"] = round(tflops / 78.6, 4)
        if tflops / 78.6 > achieved / HBM_PEAK_GBS:            # the compute roof is the nearer one (codon models)
            roofline.update({"bound": "mfma", "achieved": round(tflops, 2), "peak": 78.6, "unit": "TFLOP/s",
                             "frac": round(tflops / 78.6, 4), "hbm_GBs": round(achieved, 1)})
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(bm, wl, args.cpu_sample, tl)
        out = {
            "metric": "full-tree lnL evals/sec (GTR+G4, 1e5 patterns)" if args.config == "A" else "full-tree lnL evals/sec",
            "value": round(evals_per_s, 3), "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d taxa x %d unique patterns, %d states, %d rate categories, %s tree (%d dependency levels), "
                                   "DYNAMIC rescaling steady state" % (wl.name, wl.tip_count, wl.pattern_count, wl.state_count,
                                                                     wl.category_count, args.tree, wl.tree.depth()),
                       "patterns_per_gpu": shard.pattern_count, "parallelism": "pattern-shard x%d + 1 all-reduce" % world,
                       "ops_per_eval": int(counters["last_op_count"]), "matrices_per_eval": int(counters["last_branch_count"])},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "lnL": lnl, "lnL_first_eval": lnl0, "hbm_bytes_resident": int(raw.deviceBytes()),
            "workload_generation_s": round(t_gen, 1),
        }
        print(json.dumps(out), flush=True)
    tl.close()
    if dist is not None:
        dist.destroy_process_group()
    return out


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


def cpu_baseline(bm, wl, sample, gpu_tl):
    """The CPU oracle (oracle/beagle_cpu_oracle.c — a plain-C restatement, NOT beagle-lib) timed on this box's
    host cores on a bounded sample of the same workload: the full tree, `sample` of the P patterns (patterns
    are independent, so cost is linear in P).  Reported scaled to the full pattern count.  Also cross-checks the
    GPU's site log-likelihoods for the sampled patterns."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from beast_mcmc_amd.inputs.synth import Workload
    from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood
    n = min(sample, wl.pattern_count)
    idx = np.sort(np.random.default_rng(0).choice(wl.pattern_count, size=n, replace=False))
    sub = Workload(wl.name + "-sample", wl.tree, wl.eig, wl.freqs, wl.cat_rates, wl.cat_weights,
                   np.ascontiguousarray(wl.tip_states[:, idx]), wl.weights[idx], wl.state_count)
    lib = helpers.oracle_library()
    # a container's CPU quota can be far below the core count OpenMP sees: helpers.oracle_library() caps the oracle's
    # thread count to helpers.granted_cpus(), so the baseline runs on what the box actually grants
    threads = lib.lib.oracle_threads()
    from beast_mcmc_amd.treelikelihood import RESCALE_DYNAMIC
    o = BeagleTreeLikelihood(sub, library=lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    o.getLogLikelihood()
    o.makeDirty(); o.getLogLikelihood()                       # steady state (read-mode scalers), as on the GPU
    reps, t0 = 0, time.perf_counter()
    while True:
        o.set_substitution_model(wl.eig, wl.freqs)
        o.getLogLikelihood()
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or reps >= 200:
            break
    site_cpu = o.getSiteLogLikelihoods()
    site_gpu = gpu_tl.getSiteLogLikelihoods()[idx]
    rel = float(np.max(np.abs(site_gpu - site_cpu) / np.abs(site_cpu)))
    o.close()
    sample_evals_per_s = reps / dt
    # the same port on ONE core (SURVEY 8d asks for both): a tenth of the sample, at most ~10 s
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
            if dt1 > 8.0 or reps1 >= 50:
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


if __name__ == "__main__":
    main()
