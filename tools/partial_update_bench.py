"""Cost of the evaluations an MCMC chain issues MOST: one branch changes, the path to the root is recomputed
(BeagleTreeLikelihood.java:863-1113 with a few dirty nodes), half of the proposals rejected (restoreState).
Config A tree and alignment; prints microseconds per evaluation.  The knob of interest is BEAGLE_MI355_VSTEPS (size of
the virtual subtrees, planner.h): larger definitions make full evaluations cheaper (fewer stored nodes) and every
partial update re-evaluates the virtual siblings it passes.
Run on the GPU box:  python tools/partial_update_bench.py [patterns] [moves]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module

synth = import_module("beast-mcmc_amd.inputs.synth")
tlm = import_module("beast-mcmc_amd.treelikelihood")

patterns = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
moves = int(sys.argv[2]) if len(sys.argv) > 2 else 400
wl = synth.config_a()
if patterns < wl.pattern_count:
    wl = wl.shard(0, patterns)
tl = tlm.BeagleTreeLikelihood(wl, resource_list=[1], rescaling=tlm.RESCALE_DYNAMIC, delay_rescaling=False)
tl.getLogLikelihood()
tl.makeDirty()
t0 = time.perf_counter()
full = tl.getLogLikelihood()
t_full = time.perf_counter() - t0
rng = np.random.default_rng(5)
T, N = wl.tree.tip_count, wl.tree.node_count
height = np.array(wl.tree.height, dtype=float)
for warm in (True, False):
    n = 40 if warm else moves
    t0 = time.perf_counter()
    for i in range(n):
        node = int(rng.integers(T, N))
        while wl.tree.parent[node] < 0:
            node = int(rng.integers(T, N))
        # a new height between the node's children and its parent (a negative branch length is an error, as in BEAST)
        lo = max(height[int(wl.tree.left[node])], height[int(wl.tree.right[node])])
        hi = height[wl.tree.parent[node]]
        old = float(height[node])
        tl.storeState()
        height[node] = lo + (hi - lo) * float(rng.uniform(0.1, 0.9))
        tl.set_node_height(node, float(height[node]))
        v = tl.getLogLikelihood()
        if rng.random() < 0.5:
            tl.restoreState()
            tl.restore_node_height(node, old)
            height[node] = old
            tl.getLogLikelihood()
    dt = time.perf_counter() - t0
print("patterns %d  VSTEPS %s: full evaluation %.0f us; branch move (proposal + 50%% restore) %.1f us per move; lnL %.6f"
      % (patterns, os.environ.get("BEAGLE_MI355_VSTEPS", "default"), 1e6 * t_full, 1e6 * dt / moves, v))
tl.close()
