"""MI355X-native tree-likelihood engine behind BEAST's ``beagle.Beagle`` surface.

Layout (only what the hot path needs — SURVEY.md §8):
  csrc/     hand-written gfx950 HIP kernels, the C ABI (include/beagle_mi355.h) and the JNI shim
  (the C++ mirror of the reference's BEAGLE caller — BeagleTreeLikelihood / BufferIndexHelper — is harness, not package:
   tools/host/tree_likelihood.cpp, built into lib/libbeast_host.so)
  beagle.py           ctypes binding with the ``beagle.Beagle`` method set
  treelikelihood.py   ctypes handle on the C++ host driver
  inputs/   what feeds the engine: eigen systems, gamma rate categories, site patterns, trees, synthetic workloads
  sharding.py         pattern-block sharding across GPUs + the single lnL all-reduce

The directory name contains a hyphen, so import it through the root-level shim: ``import beast_mcmc_amd``.
"""
from . import beagle, multipartition, treelikelihood          # noqa: F401
from .inputs import patterns, siterates, substmodel, synth, trees   # noqa: F401
