"""The reference's AS 91 / AS 32 gamma quantile now lives in the package (beast_mcmc_amd.inputs.as91_quantile, the
front-end's default); the golden-value tests keep importing it under this name."""
from beast_mcmc_amd.inputs.as91_quantile import *          # noqa: F401,F403
from beast_mcmc_amd.inputs.as91_quantile import gamma_quantile   # noqa: F401
