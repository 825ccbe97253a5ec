#!/usr/bin/env python3
"""Where does a step of the pattern-sharded path spend its host time?  Run under torch.distributed.run:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
        tools/shard_overhead.py --patterns 12500 --mode shard

modes:  plain       single-GPU path (host-synchronous calculateRootLogLikelihoods), no process group
        plain_pg    the same, but with the NCCL process group initialised (and unused)
        shard_noar  sharded path on a dedicated torch stream, all_reduce skipped
        shard       sharded path as bench.py runs it
Prints the mean step time and, for the sharded modes, a per-phase breakdown (rank 0)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patterns", type=int, default=12500)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--mode", default="shard")
    ap.add_argument("--timer", action="store_true", help="plain modes: switch the engine's HIP-event kernel timer on")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import beast_mcmc_amd as bm
    from beast_mcmc_amd.sharding import ShardedTreeLikelihood
    from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_DYNAMIC
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_pg = args.mode != "plain"
    if use_pg:
        dist.init_process_group(backend="nccl", device_id=device)
        x = torch.zeros(1, device=device); dist.all_reduce(x); torch.cuda.synchronize()     # force communicator creation
    os.makedirs("/tmp/beagle_mi355_cache", exist_ok=True)
    wl = bm.synth.cached("/tmp/beagle_mi355_cache/wl_A_1_coalescent.pkl", lambda: bm.synth.config_a())
    wl = wl.shard(0, args.patterns * world)
    kw = dict(resource_list=(local + 1,), rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    if args.mode.startswith("plain"):
        tl = BeagleTreeLikelihood(wl, **kw)
        for _ in range(5):
            tl.storeState(); tl.set_substitution_model(wl.eig, wl.freqs); tl.getLogLikelihood()
        torch.cuda.synchronize()
        if args.timer:
            raw = bm.beagle.Beagle.__new__(bm.beagle.Beagle)
            raw.lib, raw._f, raw.instance = tl.engine, tl.engine.fn, tl.instance
            raw.kernelTimer(True)
        t0 = time.perf_counter()
        per = []
        for _ in range(args.steps):
            a = time.perf_counter()
            tl.storeState(); tl.set_substitution_model(wl.eig, wl.freqs); tl.set_site_model(wl.cat_rates, wl.cat_weights)
            tl.getLogLikelihood()
            per.append(int(1e6 * (time.perf_counter() - a)))
        dt = time.perf_counter() - t0
        print("per-step us:", per)
        print("%-10s patterns %d  step %.1f us  (threads in process: %d)" % (args.mode, args.patterns, 1e6 * dt / args.steps,
                                                                         len(os.listdir("/proc/self/task"))))
        tl.close()
    else:
        tl = ShardedTreeLikelihood(wl, rank, world, dist=dist, device=device, **kw)
        loc = tl.local
        for _ in range(5):
            loc.storeState(); loc.set_substitution_model(wl.eig, wl.freqs); tl.getLogLikelihood()
        names = ["storeState", "set_subst_model", "set_site_model", "prepare", "attempt_device", "all_reduce(call)", "read-back", "finish"]
        acc = [0.0] * len(names)
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        for _ in range(args.steps):
            t = [time.perf_counter()]
            loc.storeState(); t.append(time.perf_counter())
            loc.set_substitution_model(wl.eig, wl.freqs); t.append(time.perf_counter())
            loc.set_site_model(wl.cat_rates, wl.cat_weights); t.append(time.perf_counter())
            loc.prepare(); t.append(time.perf_counter())
            with tl.stream_ctx():
                loc.attempt_device(tl._buf.data_ptr()); t.append(time.perf_counter())
                if args.mode == "shard":
                    dist.all_reduce(tl._buf)
                t.append(time.perf_counter())
                tl._host.copy_(tl._buf, non_blocking=True)
            tl._stream.synchronize()
            v = float(tl._host[0]); t.append(time.perf_counter())
            loc.finish(v); t.append(time.perf_counter())
            for i in range(len(names)):
                acc[i] += t[i + 1] - t[i]
        t_all = time.perf_counter() - t_all
        if rank == 0:
            print("%-10s patterns/rank %d ranks %d  step %.1f us  (threads in process: %d)"
                  % (args.mode, args.patterns, world, 1e6 * t_all / args.steps, len(os.listdir("/proc/self/task"))))
            print("   " + "  ".join("%s %.0f" % (n, 1e6 * a / args.steps) for n, a in zip(names, acc)))
        tl.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
