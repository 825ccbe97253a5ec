import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import beast_mcmc_amd as bm, helpers
from beast_mcmc_amd.treelikelihood import *
C, T, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
wl = helpers.random_workload(T, P, 4, C, seed=900 + C + T, tree_kind="coalescent")
res = []
for fast in (1, 0):
    os.environ["BEAGLE_MI355_NO_FAST_WALK"] = "0" if fast else "1"
    tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_ALWAYS, delay_rescaling=False, traversal=POST_ORDER)
    raw = bm.beagle.Beagle.attach(tl)
    l = tl.getLogLikelihood()
    nodes = list(range(wl.tree.tip_count, wl.tree.node_count))
    sc = [raw.getLogScaleFactors(tl.node_scale_index(n)).copy() for n in nodes]
    pa = [raw.getPartials(tl.node_buffer_index(n), bm.beagle.NONE).copy() for n in nodes]
    res.append((l, sc, pa))
    tl.close()
print("lnL", res[0][0], res[1][0])
for i, n in enumerate(nodes):
    a, b = res[0][1][i], res[1][1][i]
    if not np.array_equal(a, b):
        bad = np.nonzero(a != b)[0]
        print("node", n, "scale mismatch at", len(bad), "patterns; first", bad[:8], a[bad[:4]], b[bad[:4]])
        break
for i, n in enumerate(nodes):
    a, b = res[0][2][i], res[1][2][i]
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        print("node", n, "partials mismatch at", len(bad), "entries; first", bad[:6].tolist(), a[tuple(bad[0])], b[tuple(bad[0])])
        break
