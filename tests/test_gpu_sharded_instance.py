"""The pattern-sharded multi-GPU instance INSIDE the library (beast-mcmc_amd/csrc/sharded.cpp): resource G+1 = "all GPUs".

The caller drives ONE handle exactly as it drives a single-GPU instance; the library splits the patterns into contiguous
blocks with BEAST's block sizes (src/dr/evolution/alignment/Patterns.java:142-167), runs one engine instance per block and
all-reduces the per-shard root log-likelihoods over RCCL.  On the one-GPU test box:
  * resource G+1 with one shard takes the RCCL path end to end (communicator of size 1, ncclAllReduce on the shard's stream);
  * BEAGLE_MI355_SHARDS=n places n shards on the one device, which exercises every split / gather / sum of the routing
    layer (RCCL needs distinct devices, so that configuration adds the partial sums on the host).
With >= 2 GPUs visible the second test also runs one shard per GPU over RCCL (skipped otherwise)."""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.inputs import synth
from beast_mcmc_amd.multipartition import MultiPartitionTreeLikelihood
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC, RESCALE_NONE

pytestmark = pytest.mark.gpu


def gpu_count():
    return len(bm.beagle.engine().resource_list()) - 2          # [CPU placeholder, GPU 1..G, all-GPUs]


@pytest.fixture
def shards(request):
    n = getattr(request, "param", 0)
    old = os.environ.get("BEAGLE_MI355_SHARDS")
    if n:
        os.environ["BEAGLE_MI355_SHARDS"] = str(n)
    yield n
    if old is None:
        os.environ.pop("BEAGLE_MI355_SHARDS", None)
    else:
        os.environ["BEAGLE_MI355_SHARDS"] = old


def test_resource_list_names_the_sharded_resource():
    rl = bm.beagle.engine().resource_list()
    assert "sharded" in rl[-1][0] and "RCCL" in rl[-1][1]


@pytest.mark.parametrize("shards", [0, 3, 7], indirect=True)
@pytest.mark.parametrize("rescaling", [RESCALE_NONE, RESCALE_DYNAMIC])
def test_sharded_instance_equals_single_instance(shards, rescaling, oracle_lib):
    g = gpu_count()
    wl = helpers.random_workload(40, 3001, 4, 4, seed=901)
    single = BeagleTreeLikelihood(wl, rescaling=rescaling, delay_rescaling=False)
    multi = BeagleTreeLikelihood(wl, rescaling=rescaling, delay_rescaling=False, resource_list=(g + 1,))
    for step in range(3):                       # write-mode evaluation, read mode, a branch-rate move with flipped buffers
        if step == 2:
            for t in (single, multi):
                t.storeState()
                t.set_branch_rates(np.full(wl.tree.node_count, 1.07))
        a, b = single.getLogLikelihood(), multi.getLogLikelihood()
        assert helpers.rel_err(b, a) <= 1e-12, (step, a, b)
        assert np.array_equal(single.getSiteLogLikelihoods(), multi.getSiteLogLikelihoods())      # gathered, bitwise
        single.makeDirty(); multi.makeDirty()
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=rescaling, delay_rescaling=False)
    o.storeState(); o.set_branch_rates(np.full(wl.tree.node_count, 1.07))
    assert helpers.rel_err(multi.getLogLikelihood(), o.getLogLikelihood()) <= 1e-10
    o.close(); single.close(); multi.close()


@pytest.mark.parametrize("shards", [0, 3], indirect=True)
def test_sharded_branch_gradient(shards, oracle_lib):
    """The gradient pass on the sharded handle (SURVEY 8f row f1 over 8e): every shard holds its pre-order list back and answers
    the derivative sums from it; the library adds the shards' sums.  A chain of three evaluations with alternating buffers,
    the last one with second derivatives, against the oracle."""
    from beast_mcmc_amd.gradient import BranchGradient
    g = gpu_count()
    wl = helpers.random_workload(21, 1500, 4, 4, seed=77)
    m = BranchGradient(wl, resource_list=(g + 1,), double_buffer=True)
    o = BranchGradient(wl, library=oracle_lib, double_buffer=True)
    for step in range(3):
        m.branch_lengths *= 1.1; o.branch_lengths *= 1.1
        rm, ro = m.gradient(second=step == 2), o.gradient(second=step == 2)
        assert helpers.rel_err(rm[0], ro[0]) <= 1e-10
        for a, b in zip(rm[1:], ro[1:]):
            assert np.max(np.abs(a - b)) <= 1e-10 * max(1.0, np.max(np.abs(b))), step
    assert m.b.gradientStats() == {"fused": 2, "by_operation": 0, "walked": 1, "late": 0}       # (shard 0's counters: first step and the one with squares)
    m.close(); o.close()


@pytest.mark.parametrize("shards", [4], indirect=True)
def test_sharded_gather_of_partials_scale_factors_and_20_states(shards):
    g = gpu_count()
    for S in (4, 20):
        wl = helpers.random_workload(12, 517, S, 3, seed=77 + S)
        a = BeagleTreeLikelihood(wl, rescaling=RESCALE_ALWAYS, delay_rescaling=False)
        b = BeagleTreeLikelihood(wl, rescaling=RESCALE_ALWAYS, delay_rescaling=False, resource_list=(g + 1,))
        assert helpers.rel_err(b.getLogLikelihood(), a.getLogLikelihood()) <= 1e-12
        ra, rb = bm.beagle.Beagle.attach(a), bm.beagle.Beagle.attach(b)
        for node in range(wl.tip_count, wl.tree.node_count):
            pa = ra.getPartials(a.node_buffer_index(node), bm.beagle.NONE)
            pb = rb.getPartials(b.node_buffer_index(node), bm.beagle.NONE)
            assert np.array_equal(pa, pb), (S, node)
            assert np.array_equal(ra.getLogScaleFactors(a.node_scale_index(node)), rb.getLogScaleFactors(b.node_scale_index(node)))
        a.close(); b.close()


@pytest.mark.parametrize("shards", [0, 3], indirect=True)
def test_sharded_partitioned_instance(shards):
    """calculateRootLogLikelihoodsByPartition: the all-reduce carries partitionCount doubles; a shard may hold no pattern
    of some partition."""
    g = gpu_count()
    pw = synth.config_e(scale=0.05)
    one = MultiPartitionTreeLikelihood(pw)
    many = MultiPartitionTreeLikelihood(pw, resource_list=(g + 1,))
    a, ta = one.calculate()
    b, tb = many.calculate()
    assert np.max(np.abs(a - b) / np.abs(a)) <= 1e-12 and helpers.rel_err(tb, ta) <= 1e-12
    assert np.array_equal(one.getSiteLogLikelihoods(), many.getSiteLogLikelihoods())
    one.close(); many.close()


def test_one_shard_per_gpu_over_rccl():
    g = gpu_count()
    if g < 2:
        pytest.skip("needs >= 2 GPUs in one process (the driver's multi-GPU node)")
    wl = helpers.random_workload(60, 20000, 4, 4, seed=31)
    a = BeagleTreeLikelihood(wl)
    b = BeagleTreeLikelihood(wl, resource_list=(g + 1,))
    assert helpers.rel_err(b.getLogLikelihood(), a.getLogLikelihood()) <= 1e-12
    a.close(); b.close()
