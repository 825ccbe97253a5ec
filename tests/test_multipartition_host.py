"""Host logic of the partitioned caller (beast_mcmc_amd/multipartition.py) on the CPU oracle: buffer-offset bookkeeping of full
evaluations, node-height moves and their rejection (MultiPartitionDataLikelihoodDelegate's call sequence with BufferIndexHelper
flips).  The oracle restates no pattern-partition call, so the harness runs with ONE partition here — the sequence is the same,
updatePartials instead of updatePartialsByPartition; real partitions are covered on the GPU (tests/test_gpu_configs.py)."""
import numpy as np

import helpers
from beast_mcmc_amd.inputs import synth
from beast_mcmc_amd.inputs.synth import PartitionedWorkload
from beast_mcmc_amd.multipartition import MultiPartitionTreeLikelihood
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_NONE


def fresh(w, lib):
    o = BeagleTreeLikelihood(w, library=lib, rescaling=RESCALE_NONE, delay_rescaling=False)
    v = o.getLogLikelihood()
    o.close()
    return v


def test_moves_and_rejections_on_one_partition(oracle_lib):
    pw0 = synth.config_e(scale=0.02)
    pw = PartitionedWorkload("one partition", pw0.tree, pw0.parts[2:3])
    tree = pw.tree
    tl = MultiPartitionTreeLikelihood(pw, library=oracle_lib)
    by, total = tl.calculate()
    assert helpers.rel_err(total, fresh(pw.parts[0], oracle_lib)) <= 1e-13 and by.shape == (1,)
    rng = np.random.default_rng(3)
    accepted = []
    for it in range(12):
        node = tree.root if it == 4 else int(rng.integers(tree.tip_count, tree.node_count))
        lo = max(tree.height[int(tree.left[node])], tree.height[int(tree.right[node])])
        hi = tree.height[tree.parent[node]] if node != tree.root else tree.height[node] * 1.2
        before = (float(tree.height[node]), tl.flip.copy(), tl.mf.copy())
        _, v = tl.move_node_height(node, lo + (hi - lo) * float(rng.uniform(0.1, 0.9)))
        assert helpers.rel_err(v, fresh(pw.parts[0], oracle_lib)) <= 1e-13, (it, node)      # the moved tree, evaluated from scratch
        if it % 2:
            tl.restore_move()
            assert tree.height[node] == before[0] and np.array_equal(tl.flip, before[1]) and np.array_equal(tl.mf, before[2])
        else:
            accepted.append(node)
    assert accepted
    # the accepted state in full, on the instance that moved (all offsets flip together again) and from scratch
    _, again = tl.calculate()
    assert helpers.rel_err(again, fresh(pw.parts[0], oracle_lib)) <= 1e-13
    tl.close()
