"""Where one evaluation of config E (four partitions on one instance) spends host time, call by call.
Run on the GPU box:  python tools/step_profile_E.py"""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import beast_mcmc_amd as bm
from beast_mcmc_amd.multipartition import MultiPartitionTreeLikelihood

pw = bm.synth.config_e()
tl = MultiPartitionTreeLikelihood(pw)
acc = collections.defaultdict(float)
b = tl.b
for name in ("setEigenDecomposition", "setCategoryRatesWithIndex", "updateTransitionMatricesWithMultipleModels", "updatePartialsByPartition",
             "setCategoryWeights", "setStateFrequencies", "calculateRootLogLikelihoodsByPartition"):
    f = getattr(b, name)
    def wrap(f=f, name=name):
        def g(*a, **k):
            t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; return r
        return g
    setattr(b, name, wrap())
rates0 = np.ones(pw.tree.node_count)
for i in range(20):
    tl.set_branch_rates(rates0 * (1 + 1e-6 * (i & 1))); tl.calculate()
acc.clear()
N = 100
t0 = time.perf_counter()
for i in range(N):
    tl.set_branch_rates(rates0 * (1 + 1e-6 * (i & 1))); tl.calculate()
total = time.perf_counter() - t0
print("step %.1f us" % (1e6 * total / N))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-48s %8.1f us" % (k, 1e6 * v / N))
print("  %-48s %8.1f us" % ("(python outside Beagle calls)", 1e6 * (total - sum(acc.values())) / N))
tl.close()
