"""round 6: config E, a few evaluations, how its by-partition root calls were run (walkLaunchInfo) and the host timing of the engine."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import beast_mcmc_amd as bm
from beast_mcmc_amd.inputs import synth
from beast_mcmc_amd.multipartition import MultiPartitionTreeLikelihood
pw = synth.config_e()
tl = MultiPartitionTreeLikelihood(pw)
r0 = np.ones(pw.tree.node_count)
for i in range(30):
    tl.set_branch_rates(r0 * (1.0 + 1e-6 * (i & 1)))
    by, tot = tl.calculate()
t = time.perf_counter()
for i in range(300):
    tl.set_branch_rates(r0 * (1.0 + 1e-6 * (i & 1)))
    by, tot = tl.calculate()
dt = time.perf_counter() - t
print("ms/step", 1e3 * dt / 300, "lnL", repr(tot), tl.b.walkLaunchInfo())
tl.close()
