"""Where a gradient evaluation's wall-clock time goes, call by call of the beagle.Beagle surface (host time of each call; calls that
wait for the device — the root likelihood, the edge sums — carry the kernels in front of them).  python tools/gradient_profile.py [patterns]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import beast_mcmc_amd as bm                                  # noqa: E402
from beast_mcmc_amd.gradient import BranchGradient           # noqa: E402
from beast_mcmc_amd.inputs import synth                      # noqa: E402

patterns = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
wl = synth.cached("/tmp/beagle_mi355_cache/config_A.pkl", synth.config_a) if os.path.isdir("/tmp/beagle_mi355_cache") else synth.config_a()
wl = wl.shard(0, min(patterns, wl.pattern_count))
g = BranchGradient(wl, double_buffer=True)
for _ in range(4):
    g.gradient()
g.b.synchronize()
acc = {}
cls = type(g.b)
for name in ("updateTransitionMatrices", "updatePartials", "calculateRootLogLikelihoods", "setPartials", "updatePrePartials", "setDifferentialMatrix",
             "calculateEdgeDifferentials", "resetScaleFactors", "accumulateScaleFactors", "setTransitionMatrix"):
    if not hasattr(cls, name):
        continue
    orig = getattr(cls, name)

    def wrap(self, *a, _o=orig, _n=name, **k):
        t = time.perf_counter()
        r = _o(self, *a, **k)
        acc[_n] = acc.get(_n, 0.0) + time.perf_counter() - t
        return r
    setattr(cls, name, wrap)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    g.gradient()
g.b.synchronize()
total = (time.perf_counter() - t0) / n
print("gradient at %d patterns: %.3f ms" % (wl.pattern_count, total * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-28s %8.1f us" % (k, v / n * 1e6))
print("  %-28s %8.1f us" % ("(python between the calls)", (total - sum(acc.values()) / n) * 1e6))
