"""The reference's AS 91 / AS 32 gamma quantile (tests/reference_quantile_impl.py: a transliteration of
GammaDistribution.java:530-604 / GammaFunction.java:49-198 — HARNESS, kept out of the package: it exists so that the golden
values, which the reference computed with its own approximation, can be reproduced to their last digit, and so that synthetic
workloads carry the category rates a BEAST run would hand the engine).  beast_mcmc_amd.inputs.siterates loads it from here
when ``quantile="beast"`` is asked for."""
from reference_quantile_impl import *          # noqa: F401,F403
from reference_quantile_impl import gamma_quantile   # noqa: F401
