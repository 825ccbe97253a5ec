"""Kernel time of one full evaluation of a config's operation list, whatever the values are (timing experiments with builds whose
results are wrong by construction: no rescaling protocol, so no retry).  python tools/walk_time.py [patterns] [A|B|C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import beast_mcmc_amd as bm
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_NONE
pats = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cfg = sys.argv[2] if len(sys.argv) > 2 else "A"
wl = bm.synth.cached("/tmp/wl_%s.pkl" % cfg, {"A": bm.synth.config_a, "B": bm.synth.config_b, "C": bm.synth.config_c}[cfg])
if pats < wl.pattern_count:
    wl = wl.shard(0, pats)
tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_NONE, delay_rescaling=False)
raw = bm.beagle.Beagle.attach(tl)
for i in range(5):
    tl.makeDirty(); tl.getLogLikelihood()
raw.kernelTimer(True)
N = 40 if cfg == 'A' else 10
for i in range(N):
    tl.makeDirty(); tl.getLogLikelihood()
ms, launches = raw.kernelTimer(False)
print("kernel us per evaluation: %.1f" % (1e3 * ms / N))
