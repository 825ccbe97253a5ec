#!/usr/bin/env python3
"""what an ancestral-state / Markov-jump logger pays to read partials back.

AncestralStateBeagleTreeLikelihood.traverseSample reads the partials of EVERY internal node once per logged state
(src/dr/evomodel/treelikelihood/AncestralStateBeagleTreeLikelihood.java:414-542) and the branch matrices one by one (:331).
Measures, on the GPU box: getPartials node by node (the unchanged Java caller), beagleMi355GetPartialsBatch (all internal
nodes in one call) and getTransitionMatrix per branch.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                       # noqa: E402
import beast_mcmc_amd as bm              # noqa: E402
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_DYNAMIC   # noqa: E402


def main():
    out = {}
    for name, wl in (("A/10: 1000 taxa x 10000 patterns, 4 states", bm.synth.config_a(scale=1.0).shard(0, 10000)),
                     ("B/10: 500 taxa x 5000 patterns, 20 states", bm.synth.config_b(scale=1.0).shard(0, 5000))):
        tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
        tl.getLogLikelihood()
        raw = bm.beagle.Beagle.attach(tl)
        nodes = list(range(wl.tip_count, wl.tree.node_count))
        bufs = [tl.node_buffer_index(n) for n in nodes]
        scales = [tl.node_scale_index(n) for n in nodes]
        per = wl.category_count * wl.pattern_count * wl.state_count * 8
        raw.getPartials(bufs[0], scales[0])
        t0 = time.perf_counter()
        for b, s in zip(bufs, scales):
            raw.getPartials(b, s)
        t_each = time.perf_counter() - t0
        tl.makeDirty(); tl.getLogLikelihood()              # virtual buffers are definitions again
        t0 = time.perf_counter()
        batch = raw.getPartialsBatch(bufs, scales)
        t_batch = time.perf_counter() - t0
        assert np.array_equal(batch[5], raw.getPartials(bufs[5], scales[5]))
        t0 = time.perf_counter()
        for m in range(2 * wl.tip_count - 2):
            raw.getTransitionMatrix(m)
        t_mat = time.perf_counter() - t0
        out[name] = {"internal_nodes": len(nodes), "MB_per_node": round(per / 1e6, 2),
                     "getPartials_per_node_ms": round(1e3 * t_each / len(nodes), 3), "getPartials_GBps": round(per * len(nodes) / t_each / 1e9, 2),
                     "batch_total_ms": round(1e3 * t_batch, 1), "batch_GBps": round(per * len(nodes) / t_batch / 1e9, 2),
                     "getTransitionMatrix_us_each": round(1e6 * t_mat / (2 * wl.tip_count - 2), 1)}
        tl.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
