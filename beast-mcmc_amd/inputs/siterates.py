"""Among-site rate categories handed to ``setCategoryRates`` / ``setCategoryWeights``.

Mirror of src/dr/evomodel/siteratemodel/GammaSiteRateModel.java:233-268 (``calculateCategoryRates``) and :445-472
(``setEqualRates`` / ``normalize``): equal-probability discretisation at the class medians.  The gamma quantile is a
parameter: ``quantile="exact"`` (the DEFAULT) is the exact inverse CDF (scipy, imported when first used); any callable
``(y, shape, scale) -> x`` can be passed instead.  The reference itself uses an AS 91 / AS 32 approximation that is within
~1e-6 of the exact quantile (GammaDistribution.java:530-604); its transliteration is test harness
(tests/reference_quantile_impl.py) and is handed in explicitly — ``quantile=reference_quantile.gamma_quantile`` — by the tests
that reproduce the reference's golden values to their last digit.  This package does not depend on the test tree.
"""


def exact_gamma_quantile(y, shape, scale):
    from scipy.stats import gamma                # lazily: the package does not depend on scipy otherwise
    return float(gamma.ppf(y, shape, scale=scale))


def resolve_quantile(quantile):
    if quantile is None or quantile == "exact":
        return exact_gamma_quantile
    if callable(quantile):
        return quantile
    raise ValueError("quantile must be 'exact' or a callable (y, shape, scale) -> x, not %r" % (quantile,))


class GammaSiteRateModel:
    """Mirror of dr.evomodel.siteratemodel.GammaSiteRateModel (equal-probability discretisation).

    ``alpha=None`` -> no gamma; ``p_inv=None`` -> no invariant class; ``gamma_categories`` is the
    number of gamma classes (the reference's categoryCount minus the invariant class).
    """

    def __init__(self, alpha=None, gamma_categories=1, p_inv=None, mu=1.0, quantile=None):
        self.quantile = resolve_quantile(quantile)
        self.alpha = alpha
        self.p_inv = p_inv
        self.gamma_categories = gamma_categories if alpha is not None else (1 if p_inv is not None else 1)
        self.mu = mu

    @property
    def category_count(self):
        if self.alpha is not None:
            return self.gamma_categories + (1 if self.p_inv is not None else 0)
        return 2 if self.p_inv is not None else 1

    def category_rates_and_proportions(self):
        n = self.category_count
        rates = [0.0] * n
        props = [0.0] * n
        offset = 0
        if self.p_inv is not None:
            rates[0] = 0.0
            props[0] = self.p_inv
            offset = 1
        if self.alpha is not None:
            k = n - offset
            for i in range(k):
                rates[i + offset] = self.quantile((2.0 * i + 1.0) / (2.0 * k), self.alpha, 1.0 / self.alpha)
                props[i + offset] = 1.0
            # normalize(): unweighted mean over ALL classes, proportions to sum 1
            mean = sum(rates) / n
            tot = sum(props)
            rates = [r / mean for r in rates]
            props = [p / tot for p in props]
        elif offset > 0:
            rates[offset] = 2.0
            props[offset] = 1.0 - props[0]
        else:
            rates[0] = 1.0
            props[0] = 1.0
        rates = [r * self.mu for r in rates]
        return rates, props


class OldGammaSiteModel:
    """Mirror of the OLDER discretisation used by the pure-Java ``TreeLikelihood`` tests
    (src/dr/oldevomodel/sitemodel/GammaSiteModel.java:271-311): the textbook Gamma+I mixture —
    gamma classes share ``1 - pInv`` equally and are normalised so the overall mean rate is 1."""

    def __init__(self, alpha=None, gamma_categories=1, p_inv=None, quantile=None):
        self.alpha, self.gamma_categories, self.p_inv = alpha, gamma_categories, p_inv
        self.quantile = resolve_quantile(quantile)

    def category_rates_and_proportions(self):
        rates, props = [], []
        prop_variable = 1.0
        if self.p_inv is not None:
            rates.append(0.0)
            props.append(self.p_inv)
            prop_variable = 1.0 - self.p_inv
        if self.alpha is not None:
            k = self.gamma_categories
            q = [self.quantile((2.0 * i + 1.0) / (2.0 * k), self.alpha, 1.0 / self.alpha) for i in range(k)]
            mean = (prop_variable * sum(q)) / k
            rates += [x / mean for x in q]
            props += [prop_variable / k] * k
        else:
            rates.append(1.0 / prop_variable)
            props.append(prop_variable)
        return rates, props
