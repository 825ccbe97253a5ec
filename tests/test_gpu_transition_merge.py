"""updateTransitionMatrices merged with the matrix gather of the updatePartials call behind it (4 states).

A whole tree's updateTransitionMatrices call is held back until the next call; when that is updatePartials with a cached program
resident on the device, ONE launch computes the matrices, writes them to the caller's slots and lays them out where the walk reads
them — the program's matrix stream and the private snapshots of its virtual definitions (kernels.hip k_transition4Scatter, through
the program's inverse map matrix -> places) — instead of a transition launch and a gather launch.  Anything else that arrives
first launches the held call as it is.  Held here against BEAGLE_MI355_NO_LAUNCH_FUSION=1 (never held, separate launches): the
same BITS for log-likelihoods, site values and the matrices themselves, through model moves, height moves, rejections, matrix
read-backs between the two calls, and the counter says which route ran."""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.inputs import substmodel
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC, RESCALE_NONE

pytestmark = pytest.mark.gpu


def chain(wl, merged, scheme):
    old = os.environ.get("BEAGLE_MI355_NO_LAUNCH_FUSION")
    os.environ["BEAGLE_MI355_NO_LAUNCH_FUSION"] = "0" if merged else "1"
    try:
        tl = BeagleTreeLikelihood(wl, rescaling=scheme, delay_rescaling=False)
    finally:
        if old is None:
            os.environ.pop("BEAGLE_MI355_NO_LAUNCH_FUSION", None)
        else:
            os.environ["BEAGLE_MI355_NO_LAUNCH_FUSION"] = old
    raw = bm.beagle.Beagle.attach(tl)
    rng = np.random.default_rng(21)
    vals, mats = [tl.getLogLikelihood()], []
    for step in range(18):
        kind = step % 3
        tl.storeState()
        if kind == 0:
            tl.set_substitution_model(substmodel.gtr(rng.gamma(2.0, 1.0, size=6) + 0.1, wl.freqs), wl.freqs)
        elif kind == 1:
            tl.set_branch_rates(rng.uniform(0.5, 1.5, size=wl.tree.node_count))
        else:
            node = int(rng.integers(wl.tree.tip_count, wl.tree.node_count - 1))
            tl.set_node_height(node, helpers.proposed_height(wl.tree, node, rng))
        vals.append(tl.getLogLikelihood())
        if step % 5 == 4:
            tl.restoreState()
            vals.append(tl.getLogLikelihood())
        if step % 4 == 1:
            mats.append(raw.getTransitionMatrix(int(rng.integers(0, 2 * wl.tree.node_count))).copy())
    site = tl.getSiteLogLikelihoods().copy()
    health = raw.walkHealth()
    tl.close()
    return vals, site, mats, health


@pytest.mark.parametrize("scheme", [RESCALE_NONE, RESCALE_DYNAMIC, RESCALE_ALWAYS])
@pytest.mark.parametrize("T,P,C", [(130, 2500, 4), (70, 300, 1), (90, 1300, 8)])
def test_merged_launch_gives_the_same_bits(T, P, C, scheme, oracle_lib):
    wl = helpers.random_workload(T, P, 4, C, seed=9100 + T)
    mv, ms, mm, mh = chain(wl, True, scheme)
    uv, us, um, uh = chain(wl, False, scheme)
    assert mh["merged_transition_launches"] > 0 and uh["merged_transition_launches"] == 0
    assert mv == uv
    assert np.array_equal(ms, us)
    for a, b in zip(mm, um):
        assert np.array_equal(a, b)
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=scheme, delay_rescaling=False)
    assert helpers.rel_err(mv[0], o.getLogLikelihood()) <= 1e-10
    o.close()


def test_calls_between_the_two_launch_the_held_call_as_it_is():
    """Raw call sequence: updateTransitionMatrices, then a read of one of its matrices (the held call has to run first), then a
    second updateTransitionMatrices of a few matrices (launched at once: a partial update's), then updatePartials."""
    wl = helpers.random_workload(80, 700, 4, 4, seed=9300)
    tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_NONE)
    ref = [tl.getLogLikelihood()]
    raw = bm.beagle.Beagle.attach(tl)
    tl.makeDirty(); ref.append(tl.getLogLikelihood())
    before = raw.walkHealth()["merged_transition_launches"]
    n = wl.tree.node_count
    idx = np.arange(0, min(70, 2 * n), dtype=np.int32)
    lens = np.linspace(0.01, 0.3, len(idx))
    raw.updateTransitionMatrices(0, idx, None, None, lens, len(idx))       # held
    m3 = raw.getTransitionMatrix(3)                                         # ... until somebody looks
    raw.updateTransitionMatrices(0, idx[:5], None, None, lens[:5] * 2.0, 5)
    m3b = raw.getTransitionMatrix(3)
    assert raw.walkHealth()["merged_transition_launches"] == before
    assert not np.array_equal(m3, m3b) and np.all(np.isfinite(m3)) and np.allclose(m3.sum(axis=2), 1.0)
    tl.close()
