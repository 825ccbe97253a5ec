#!/usr/bin/env python3
"""Where does a step of the pattern-sharded path spend its host time?  Run under torch.distributed.run (any rank count):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
        tools/shard_overhead.py --patterns 12500

Prints per-phase wall time (perf_counter, microseconds, mean over the timed steps) on rank 0."""
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
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import beast_mcmc_amd as bm
    from beast_mcmc_amd.sharding import ShardedTreeLikelihood
    from beast_mcmc_amd.treelikelihood import RESCALE_DYNAMIC
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist.init_process_group(backend="nccl", device_id=device)
    os.makedirs("/tmp/beagle_mi355_cache", exist_ok=True)
    wl = bm.synth.cached("/tmp/beagle_mi355_cache/wl_A_1_coalescent.pkl", lambda: bm.synth.config_a())
    wl = wl.shard(0, args.patterns * world)
    tl = ShardedTreeLikelihood(wl, rank, world, dist=dist, device=device, resource_list=(local + 1,),
                               rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    loc = tl.local
    for _ in range(5):
        loc.storeState(); loc.set_substitution_model(wl.eig, wl.freqs); tl.getLogLikelihood()
    names = ["store+set_model", "prepare", "attempt_device", "all_reduce(call)", "item()", "finish"]
    acc = [0.0] * len(names)
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for _ in range(args.steps):
        t = [time.perf_counter()]
        loc.storeState(); loc.set_substitution_model(wl.eig, wl.freqs); loc.set_site_model(wl.cat_rates, wl.cat_weights)
        t.append(time.perf_counter())
        loc.prepare(); t.append(time.perf_counter())
        with tl.stream_ctx():
            loc.attempt_device(tl._buf.data_ptr()); t.append(time.perf_counter())
            dist.all_reduce(tl._buf); t.append(time.perf_counter())
            v = float(tl._buf.item()); t.append(time.perf_counter())
        loc.finish(v); t.append(time.perf_counter())
        for i in range(len(names)):
            acc[i] += t[i + 1] - t[i]
    t_all = time.perf_counter() - t_all
    if rank == 0:
        print("patterns/rank %d  ranks %d  step %.1f us" % (args.patterns, world, 1e6 * t_all / args.steps))
        for n, a in zip(names, acc):
            print("  %-18s %8.1f us" % (n, 1e6 * a / args.steps))
    tl.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
