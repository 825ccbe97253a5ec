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
"""
import numpy as np

from .inputs import patterns as _patterns
from .treelikelihood import BeagleTreeLikelihood


class ShardedTreeLikelihood:
    def __init__(self, workload, rank, world_size, dist=None, device=None, library=None, **kw):
        self.rank, self.world = rank, world_size
        self.dist = dist
        start, stop = _patterns.shard_bounds(workload.pattern_count, world_size)[rank]
        self.range = (start, stop)
        self.local = BeagleTreeLikelihood(workload.shard(start, stop), library=library, **kw)
        self.device = device
        self._buf = None
        if device is not None:
            import torch
            self._torch = torch
            self._buf = torch.zeros(1, dtype=torch.float64, device=device)
            # the engine enqueues on torch's current stream so its kernels are ordered with the collective
            from . import beagle as _b
            raw = _b.Beagle.__new__(_b.Beagle)
            raw.lib, raw._f, raw.instance = self.local.engine, self.local.engine.fn, self.local.instance
            raw.setStream(torch.cuda.current_stream(device).cuda_stream)
        elif dist is not None:
            import torch
            self._torch = torch
            self._buf = torch.zeros(1, dtype=torch.float64)

    def _all_reduce(self, local_value=None):
        """Sum of the per-shard log-likelihoods over all ranks, as a Python float."""
        if self.device is not None:
            if self.dist is not None:
                self.dist.all_reduce(self._buf, op=self.dist.ReduceOp.SUM)
            return float(self._buf.item())          # the only host<-device transfer of the evaluation
        if self.dist is not None and self.world > 1:
            self._buf[0] = local_value
            self.dist.all_reduce(self._buf, op=self.dist.ReduceOp.SUM)
            return float(self._buf[0])
        return local_value

    def getLogLikelihood(self):
        tl = self.local
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
