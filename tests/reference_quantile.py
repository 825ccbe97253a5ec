"""The reference's AS 91 / AS 32 gamma quantile (tests/reference_quantile_impl.py: a transliteration of
GammaDistribution.java:530-604 / GammaFunction.java:49-198 — HARNESS, kept out of the package: it exists so that the golden
values, which the reference computed with its own approximation, can be reproduced to their last digit, and so that synthetic
workloads can carry the category rates a BEAST run would hand the engine).  Tests pass it to beast_mcmc_amd.inputs.siterates
explicitly (``quantile=reference_quantile.gamma_quantile``); the package's own default is the exact quantile."""
from reference_quantile_impl import *          # noqa: F401,F403
from reference_quantile_impl import gamma_quantile   # noqa: F401
