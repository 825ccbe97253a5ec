"""Pattern-sharded tree likelihood across the GPUs of one node: one process per GPU, unique site
patterns split into contiguous blocks, ONE all-reduce of the per-shard log-likelihood per evaluation.

Reference equivalent: ``-beagle_instances G -beagle_order 1,..,G`` — G ``BeagleDataLikelihoodDelegate``s over
``new Patterns(patterns, j, G)`` blocks (src/dr/evomodelxml/treedatalikelihood/TreeDataLikelihoodParser.java:205-236,
block sizes src/dr/evolution/alignment/Patterns.java:142-167), evaluated by ``CompoundLikelihood``'s thread pool and
summed in Java (src/dr/inference/model/CompoundLikelihood.java:202-214).  Here the sum is an RCCL all-reduce of one
fp64 over xGMI (``torch.distributed`` backend "nccl" on ROCm); with the "gloo" backend the same code runs on CPU
tensors, which is how the N>1 path is tested without GPUs.

Tree, eigen system, matrices and op lists are replicated (KB-scale); nothing else crosses GPUs.  The rescaling
retry decision is taken on the GLOBAL value, so every rank takes the same branch.

Two ways to run the collective (``collective=``):
  "engine" (default on a GPU)  the all-reduce is issued INSIDE the engine, on its own stream right behind the reduction
           kernel, through RCCL directly (include/beagle_mi355.h beagleMi355CommInit / ...AllReduce): torch.distributed only
           carries the 128-byte communicator id to the ranks once, at start-up; an evaluation is ONE call into the host driver.
  "torch"  the per-shard sum stays on the device and ``torch.distributed.all_reduce`` adds it up (what round 3 shipped; what
           the CPU tests run over gloo).

Stream discipline on the GPU (collective="torch"): the engine, the collective and the 8-byte read-back all run on ONE dedicated torch
stream (never the legacy default stream, whose handle is 0 and which the engine would read as "use your own"):
kernels -> device-side sum -> all_reduce -> D2H are ordered by the stream itself, with no event or host sync in between.
"""
import contextlib

import numpy as np

from .inputs import patterns as _patterns
from .treelikelihood import BeagleTreeLikelihood


class ShardedTreeLikelihood:
    def __init__(self, workload, rank, world_size, dist=None, device=None, library=None, collective=None, **kw):
        self.rank, self.world = rank, world_size
        self.dist = dist
        import os
        # (BEAGLE_MI355_COLLECTIVE=torch|engine: an operator's override of the default route on a GPU)
        self.collective = collective or os.environ.get("BEAGLE_MI355_COLLECTIVE") or ("engine" if device is not None else "torch")
        if self.collective not in ("engine", "torch") or device is None:
            self.collective = "torch" if device is None else ("engine" if self.collective == "engine" else "torch")
        start, stop = _patterns.shard_bounds(workload.pattern_count, world_size)[rank]
        self.range = (start, stop)
        self.local = BeagleTreeLikelihood(workload.shard(start, stop), library=library, **kw)
        self.device = device
        self._buf = None
        self._stream = None
        if device is not None and self.collective == "engine":
            import torch
            from . import beagle as _b
            raw = _b.Beagle.__new__(_b.Beagle)
            raw.lib, raw._f, raw.instance = self.local.engine, self.local.engine.fn, self.local.instance
            ok = 1
            unique = None
            if dist is not None and world_size > 1:
                # rank 0's id travels with a validity byte: a rank 0 that cannot produce one still takes part in the broadcast,
                # and every rank then skips the communicator together instead of waiting for it
                ident = torch.zeros(129, dtype=torch.uint8, device=device)
                if rank == 0:
                    try:
                        host = bytearray(raw.commUniqueId()) + bytearray([1])
                        ident.copy_(torch.frombuffer(host, dtype=torch.uint8))
                    except Exception:                 # noqa: BLE001
                        pass
                dist.broadcast(ident, src=0)
                got = bytes(ident.cpu().numpy().tobytes())
                if got[128] == 1:
                    unique = got[:128]
                else:
                    ok = 0
            else:
                try:
                    unique = raw.commUniqueId()
                except Exception:                     # noqa: BLE001
                    ok = 0
            if ok:
                # ncclCommInitRank is a rendezvous of all ranks: it runs on a helper thread with a deadline, so that a rank whose
                # peers never arrive falls back to the torch.distributed collective (with everybody else: the MIN below)
                # instead of sitting in it for ever; BEAGLE_MI355_COMM_INIT_TIMEOUT_S, default 120
                import os
                import threading
                state = {"ok": 0}

                def _init():
                    try:
                        raw.commInit(unique, rank, world_size)
                        state["ok"] = 1
                    except Exception:                 # noqa: BLE001  (decided together below)
                        state["ok"] = 0
                th = threading.Thread(target=_init, daemon=True)
                th.start()
                th.join(float(os.environ.get("BEAGLE_MI355_COMM_INIT_TIMEOUT_S", "120")))
                ok = 0 if th.is_alive() else state["ok"]
                if th.is_alive():
                    # the helper thread is still inside beagleMi355CommInit ON THAT INSTANCE — an instance is driven by one thread
                    # at a time, and a late rendezvous would set its communicator under the main thread's feet.  The instance is
                    # left to the straggler (kept alive here, never called again) and this rank continues on a fresh one.
                    self._abandoned = (self.local, th)
                    self.local = BeagleTreeLikelihood(workload.shard(start, stop), library=library, **kw)
            if dist is not None and world_size > 1:   # every rank takes the same route: all of them have a communicator, or none uses it
                flag = torch.tensor([ok], dtype=torch.int32, device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if ok:
                self.local.set_engine_collective(True)
            else:
                self.collective = "torch"
        if device is not None and self.collective != "engine":
            import torch
            self._torch = torch
            self._stream = torch.cuda.Stream(device=device)
            assert self._stream.cuda_stream != 0
            with torch.cuda.stream(self._stream):
                self._buf = torch.zeros(1, dtype=torch.float64, device=device)
                self._host = torch.zeros(1, dtype=torch.float64).pin_memory()
            from . import beagle as _b
            raw = _b.Beagle.__new__(_b.Beagle)
            raw.lib, raw._f, raw.instance = self.local.engine, self.local.engine.fn, self.local.instance
            raw.setStream(self._stream.cuda_stream)
        elif device is None and dist is not None:
            import torch
            self._torch = torch
            self._buf = torch.zeros(1, dtype=torch.float64)

    def stream_ctx(self):
        return self._torch.cuda.stream(self._stream) if self._stream is not None else contextlib.nullcontext()

    def _all_reduce(self, local_value=None):
        """Sum of the per-shard log-likelihoods over all ranks, as a Python float."""
        if self.device is not None:
            with self.stream_ctx():
                if self.dist is not None:
                    self.dist.all_reduce(self._buf, op=self.dist.ReduceOp.SUM)
                self._host.copy_(self._buf, non_blocking=True)   # the only host<-device transfer of the evaluation
            self._stream.synchronize()
            return float(self._host[0])
        if self.dist is not None and self.world > 1:
            self._buf[0] = local_value
            self.dist.all_reduce(self._buf, op=self.dist.ReduceOp.SUM)
            return float(self._buf[0])
        return local_value

    def getLogLikelihood(self):
        tl = self.local
        if self.device is not None and self.collective == "engine":
            return tl.getLogLikelihood()          # prepare / attempt (kernels, all-reduce) / finish on the global value: one call
        tl.prepare()
        while True:
            if self.device is not None:
                tl.attempt_device(self._buf.data_ptr())
                total = self._all_reduce()
            else:
                total = self._all_reduce(tl.attempt_host())
            if tl.finish(total):
                return total if np.isfinite(total) else float("-inf")

    def makeDirty(self):
        self.local.makeDirty()

    def close(self):
        self.local.close()
