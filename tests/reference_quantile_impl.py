"""The reference's gamma quantile, TRANSLITERATED statement by statement from the reference's Java (``GammaSiteRateModel(quantile="beast")``, the front-end's default).

BEAST's golden lnL values depend on the exact discretisation of the among-site rate distribution, including its gamma
quantile, which is NOT an exact inverse CDF: it is AS 91 (Best & Roberts 1975) driven by an AS 32 (Bhattacharjee 1970)
incomplete gamma with a 1e-8 stopping rule, a Pike & Hill (1966, CACM Alg. 291) log-gamma and an AS 70 (Odeh & Evans
1974) normal quantile for the starting value:

    src/dr/math/distributions/GammaDistribution.java:281-283 (quantile), :530-604 (pointChi2)
    src/dr/math/GammaFunction.java:49-68 (lnGamma), :122-198 (incompleteGamma)
    src/dr/math/ErrorFunction.java:86-120 (inverseErf / pointNormal)
    src/dr/math/distributions/NormalDistribution.java:177-179 (quantile)

Those published algorithms have to be followed statement by statement to reproduce the 1e-13 fixture of
tests/TestXML/testBranchSpecificSubstitutionModel.xml:209-235, and for a user of this front-end to get the category rates
BEAST itself would hand the engine.  ``quantile="exact"`` (scipy's inverse CDF) differs by about 2.5e-8 in absolute lnL on
that fixture."""
import math


def ln_gamma(alpha):
    # Pike & Hill (1966) Algorithm 291
    x = alpha
    f = 0.0
    if x < 7:
        f = 1.0
        z = x - 1.0
        while True:
            z += 1.0
            if not (z < 7):
                break
            f *= z
        x = z
        f = -math.log(f)
    z = 1.0 / (x * x)
    return (f + (x - 0.5) * math.log(x) - x + 0.918938533204673 +
            (((-0.000595238095238 * z + 0.000793650793651) * z - 0.002777777777778) * z +
             0.083333333333333) / x)


def incomplete_gamma(x, alpha, ln_gamma_alpha):
    # Bhattacharjee (1970) AS 32: series for x<=1 or x<alpha, continued fraction otherwise
    accurate = 1e-8
    overflow = 1e30
    if x == 0.0:
        return 0.0
    if x < 0.0 or alpha <= 0.0:
        raise ValueError("Arguments out of bounds")
    factor = math.exp(alpha * math.log(x) - x - ln_gamma_alpha)
    if x > 1 and x >= alpha:
        a = 1 - alpha
        b = a + x + 1
        term = 0.0
        pn0 = 1.0
        pn1 = x
        pn2 = x + 1
        pn3 = x * b
        gin = pn2 / pn3
        while True:
            a += 1
            b += 2
            term += 1
            an = a * term
            pn4 = b * pn2 - an * pn0
            pn5 = b * pn3 - an * pn1
            if pn5 != 0:
                rn = pn4 / pn5
                dif = abs(gin - rn)
                if dif <= accurate and dif <= accurate * rn:
                    break
                gin = rn
            pn0, pn1, pn2, pn3 = pn2, pn3, pn4, pn5
            if abs(pn4) >= overflow:
                pn0 /= overflow
                pn1 /= overflow
                pn2 /= overflow
                pn3 /= overflow
        gin = 1 - factor * gin
    else:
        gin = 1.0
        term = 1.0
        rn = alpha
        while True:
            rn += 1
            term *= x / rn
            gin += term
            if not (term > accurate):
                break
        gin *= factor / alpha
    return gin


def point_normal(prob):
    # Odeh & Evans (1974) AS 70
    a0, a1, a2, a3, a4 = -0.322232431088, -1.0, -0.342242088547, -0.0204231210245, -0.453642210148e-4
    b0, b1, b2, b3, b4 = 0.0993484626060, 0.588581570495, 0.531103462366, 0.103537752850, 0.0038560700634
    p = prob
    p1 = p if p < 0.5 else 1 - p
    y = math.sqrt(math.log(1 / (p1 * p1)))
    z = y + ((((y * a4 + a3) * y + a2) * y + a1) * y + a0) / ((((y * b4 + b3) * y + b2) * y + b1) * y + b0)
    return -z if p < 0.5 else z


def normal_quantile(z, m=0.0, sd=1.0):
    inverse_erf = point_normal(0.5 * (2.0 * z - 1.0) + 0.5) / math.sqrt(2.0)
    return m + math.sqrt(2.0) * sd * inverse_erf


def point_chi2(prob, v):
    # Best & Roberts (1975) AS 91
    e = 0.5e-6
    aa = 0.6931471805
    p = prob
    epsi = 0.01
    if p < 0.000002 or p > 1 - 0.000002:
        epsi = 0.000001
    g = ln_gamma(v / 2)
    xx = v / 2
    c = xx - 1
    if v < -1.24 * math.log(p):
        ch = math.pow(p * xx * math.exp(g + xx * aa), 1 / xx)
        if ch - e < 0:
            return ch
    else:
        if v > 0.32:
            x = normal_quantile(p, 0, 1)
            p1 = 0.222222 / v
            ch = v * math.pow(x * math.sqrt(p1) + 1 - p1, 3.0)
            if ch > 2.2 * v + 6:
                ch = -2 * (math.log(1 - p) - c * math.log(0.5 * ch) + g)
        else:
            ch = 0.4
            a = math.log(1 - p)
            while True:
                q = ch
                p1 = 1 + ch * (4.67 + ch)
                p2 = ch * (6.73 + ch * (6.66 + ch))
                t = -0.5 + (4.67 + 2 * ch) / p1 - (6.73 + ch * (13.32 + 3 * ch)) / p2
                ch -= (1 - math.exp(a + g + 0.5 * ch + c * aa) * p2 / p1) / t
                if not (abs(q / ch - 1) - epsi > 0):
                    break
    while True:
        q = ch
        p1 = 0.5 * ch
        t = incomplete_gamma(p1, xx, g)
        if t < 0:
            raise ValueError("Arguments out of range: t < 0")
        p2 = p - t
        t = p2 * math.exp(xx * aa + g + p1 - c * math.log(ch))
        b = t / ch
        a = 0.5 * t - b * c
        s1 = (210 + a * (140 + a * (105 + a * (84 + a * (70 + 60 * a))))) / 420
        s2 = (420 + a * (735 + a * (966 + a * (1141 + 1278 * a)))) / 2520
        s3 = (210 + a * (462 + a * (707 + 932 * a))) / 2520
        s4 = (252 + a * (672 + 1182 * a) + c * (294 + a * (889 + 1740 * a))) / 5040
        s5 = (84 + 264 * a + c * (175 + 606 * a)) / 2520
        s6 = (120 + c * (346 + 127 * c)) / 5040
        ch += t * (1 + 0.5 * t * s1 - b * c * (s1 - b * (s2 - b * (s3 - b * (s4 - b * (s5 - b * s6))))))
        if not (abs(q / ch - 1) > e):
            break
    return ch


def gamma_quantile(y, shape, scale):
    return 0.5 * scale * point_chi2(y, 2.0 * shape)
