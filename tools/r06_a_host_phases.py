"""round 6: config A as bench.py's main loop drives it (a new eigen system and rates every step), the caller's host time per phase
(BTL_TIMING=1: tools/host/tree_likelihood.cpp) beside the step time."""
import os, sys, time
os.environ["BTL_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import beast_mcmc_amd as bm
from beast_mcmc_amd.inputs import synth, substmodel
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_DYNAMIC
wl = synth.config_a()
tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
rng = np.random.default_rng(1)
eigs = [substmodel.gtr(rng.gamma(2.0, 1.0, size=6) + 0.1, wl.freqs) for _ in range(3)]
for i in range(30):
    tl.set_substitution_model(eigs[i % 3], wl.freqs); tl.getLogLikelihood()
tl.host_phase_times()
t = time.perf_counter()
n = 300
for i in range(n):
    tl.set_substitution_model(eigs[i % 3], wl.freqs)
    v = tl.getLogLikelihood()
dt = time.perf_counter() - t
ph = tl.host_phase_times()
print("us/step %.1f" % (1e6 * dt / n), {k: round(x / n, 1) for k, x in ph.items()} if isinstance(ph, dict) else ph, repr(v))
tl.close()
