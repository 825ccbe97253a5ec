import sys, os, threading, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import helpers
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_DYNAMIC
variant = sys.argv[1] if len(sys.argv) > 1 else "20"
second = {"20": lambda: helpers.random_workload(25, 2500, 20, 2, seed=501), "4": lambda: helpers.random_workload(30, 2000, 4, 4, seed=502),
          "61": lambda: helpers.random_workload(15, 1000, 61, 2, seed=503), "same": lambda: helpers.random_workload(40, 3000, 4, 4, seed=500)}[variant]()
wls = [helpers.random_workload(40, 3000, 4, 4, seed=500), second]
print("second instance:", variant)
SITES = len(sys.argv) > 2
sites = {}
def chain(wl, out):
    tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    vals = []
    for k in range(25):
        tl.storeState()
        tl.set_branch_rates(np.full(wl.tree.node_count, 1.0 + 0.01 * k))
        vals.append(tl.getLogLikelihood())
        if SITES: sites.setdefault(id(out), []).append(tl.getSiteLogLikelihoods().copy())
    tl.close()
    out.append(vals)
serial = []
for wl in wls: chain(wl, serial)
for rep in range(8):
    par = [[], []]
    th = [threading.Thread(target=chain, args=(wls[i], par[i])) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for i in range(2):
        bad = [k for k in range(25) if par[i][0][k] != serial[i][k]]
        for k in bad:
            same = [j for j in range(25) if par[i][0][k] == serial[i][j]]
            print("rep", rep, "instance", i, "index", k, "par", par[i][0][k], "serial", serial[i][k], "equals serial index", same)
            if SITES:
                a = sites[id(par[i])][k]; b = sites[id(serial)][k] if i == 0 else None
                if b is not None:
                    bad_p = np.nonzero(a != b)[0]
                    print("   patterns that differ:", len(bad_p), "first", bad_p[:6], "last", bad_p[-3:], "groups of 128:", sorted(set((bad_p // 128).tolist()))[:20], "max |diff|", np.max(np.abs(a - b)))
print("done")
