"""The hardware ordering the 4-state walk's waits rest on, and the switch that makes the engine independent of it.

A stage of the walk waits with ``s_waitcnt vmcnt(N)``; gfx9 counts loads and stores in one counter and the ISA guides promise
in-order return only for loads among themselves.  The DEFAULT build takes N from the next stage's loads only, which needs
nothing but that rule (engine_walk.cpp runPlan).  ``BEAGLE_MI355_STRICT_WAITS=0`` also counts the previous stage's STORES as
retiring behind the stage's own (older) loads — 1 % faster, resting on an observation: tests/native/vmcnt_order_probe.hip
looks for a counter-example in the four situations the kernels create (> 1e9 lane-trials).  Both must give the same bits."""
import os
import re
import subprocess

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_DYNAMIC

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "vmcnt_order_probe.hip")
EXE = os.path.join(ROOT, "tests", "native", "vmcnt_order_probe")


def test_younger_stores_never_retire_before_an_older_load():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", SRC, "-o", EXE])
    out = subprocess.run([EXE, "600"], capture_output=True, text=True, timeout=600)
    rows = [l.split() for l in out.stdout.splitlines() if l.startswith("PROBE")]
    assert len(rows) == 12, out.stdout + out.stderr
    trials = sum(float(r[-1]) for r in rows)
    assert trials >= 1e9
    assert all(int(r[-2]) == 0 for r in rows), out.stdout
    assert out.returncode == 0 and "in order" in out.stdout


def _evaluate(wl, strict, fast):
    env = {"BEAGLE_MI355_STRICT_WAITS": "1" if strict else "0", "BEAGLE_MI355_NO_FAST_WALK": "0" if fast else "1"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)                      # read when the instance is created
    try:
        tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    raw = bm.beagle.Beagle.attach(tl)
    first = tl.getLogLikelihood()               # write mode
    tl.makeDirty()
    lnl = tl.getLogLikelihood()                 # read mode
    site = tl.getSiteLogLikelihoods().copy()
    nodes = list(range(wl.tree.tip_count, wl.tree.node_count, 7)) + [wl.tree.node_count - 1]
    parts = [raw.getPartials(tl.node_buffer_index(n), bm.beagle.NONE).copy() for n in nodes]
    tl.close()
    return first, lnl, site, parts


@pytest.mark.parametrize("fast", [True, False])
def test_strict_waits_give_the_same_bits(fast, oracle_lib):
    """Config A's shape at a size that fills the chip several times (250 taxa x 40 000 patterns: stored nodes, re-read
    children, hold slots, slices in several waves), default waits against strict waits, on both walk kernels."""
    wl = bm.synth.config_a(scale=0.25, seed=5)
    wl = wl.shard(0, 40000) if wl.pattern_count > 40000 else wl
    d = _evaluate(wl, False, fast)
    s = _evaluate(wl, True, fast)
    assert d[0] == s[0] and d[1] == s[1]
    assert np.array_equal(d[2], s[2])
    for a, b in zip(d[3], s[3]):
        assert np.array_equal(a, b)
    # and right: the last 256 patterns and a 1 % sample against the oracle
    idx = np.unique(np.concatenate([np.arange(wl.pattern_count - 256, wl.pattern_count),
                                    np.random.default_rng(1).choice(wl.pattern_count, 400, replace=False)]))
    sub = bm.synth.Workload("sample", wl.tree, wl.eig, wl.freqs, wl.cat_rates, wl.cat_weights,
                            np.ascontiguousarray(wl.tip_states[:, idx]), wl.weights[idx], 4)
    o = BeagleTreeLikelihood(sub, library=oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    o.getLogLikelihood()
    so = o.getSiteLogLikelihoods()
    assert np.max(np.abs(s[2][idx] - so) / np.abs(so)) <= 1e-10
    o.close()
