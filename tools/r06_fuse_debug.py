"""development: node partials with fused cherries against the unfused program, node by node"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_NONE

def run(wl, fuse):
    os.environ["BEAGLE_MI355_NO_CHERRY_FUSION"] = "0" if fuse else "1"
    tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_NONE, delay_rescaling=False)
    raw = bm.beagle.Beagle.attach(tl)
    raw.kernelTimer(True)
    l = tl.getLogLikelihood()
    info = raw.walkLaunchInfo()
    nodes = list(range(wl.tree.tip_count, wl.tree.node_count))
    parts = {n: raw.getPartials(tl.node_buffer_index(n), bm.beagle.NONE).copy() for n in nodes}
    tl.close()
    return l, parts, info

T, P, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
wl = helpers.random_workload(T, P, 4, C, seed=5, tree_kind="coalescent")
lf, pf, info = run(wl, True)
lu, pu, _ = run(wl, False)
print("lnL fused", lf, "unfused", lu, info)
tr = wl.tree
for n in sorted(pf):
    a, b = pf[n].reshape(C, P, 4), pu[n].reshape(C, P, 4)
    bad = np.argwhere(a != b)
    kids = (int(tr.left[n]), int(tr.right[n]))
    print("node", n, "children", kids, "tips" if max(kids) < T else "", "mismatches", len(bad), (bad[:3].tolist(), a[tuple(bad[0])], b[tuple(bad[0])]) if len(bad) else "")
