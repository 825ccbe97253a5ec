"""Substitution-model inputs for the engine: rate matrix -> normalised eigen system.

The engine never sees a substitution model; it is handed ``(U, U^-1, lambda)`` through
``setEigenDecomposition`` (reference caller:
src/dr/evomodel/treedatalikelihood/HomogenousSubstitutionModelDelegate.java:228-240).  This module
produces that triple the way the reference's host code does:

* Q construction           src/dr/evomodel/substmodel/BaseSubstitutionModel.java:256-266 (``setupQMatrix``:
                           ``Q[i][j] = r_k * pi[j]`` over the upper-triangle rate order)
* rows sum to zero         BaseSubstitutionModel ``makeValid``
* normalisation            BaseSubstitutionModel.java:314-319 (``-sum_i Q_ii pi_i``), applied to the
                           eigenvalues (``EigenDecomposition.normalizeEigenValues``)
* GTR rate order           src/dr/evomodel/substmodel/nucleotide/GTR.java:161-184 (AC, AG, AT, CG, CT, GT)
* HKY                      src/dr/evomodel/substmodel/nucleotide/HKY.java:157-230 (kappa on AG and CT)
* GY94 codon rates         src/dr/evomodel/substmodel/codon/GY94CodonModel.java:165-188

The decomposition itself is numpy's symmetric eigensolver on the pi-symmetrised matrix (the
reference uses Colt / its own Householder-QL; any exact decomposition yields the same P(t)).
"""
import numpy as np


class EigenDecomposition:
    """Mirror of dr.evomodel.substmodel.EigenDecomposition: row-major U, U^-1 and eigenvalues."""

    def __init__(self, evec, ievc, evals):
        self.evec = np.ascontiguousarray(evec, dtype=np.float64)
        self.ievc = np.ascontiguousarray(ievc, dtype=np.float64)
        self.evals = np.ascontiguousarray(evals, dtype=np.float64)

    def transition_probabilities(self, distance):
        """BaseSubstitutionModel.java:206-245 without the abs()."""
        return (self.evec * np.exp(distance * self.evals)[None, :]) @ self.ievc


def reversible_q(rel_rates_upper, pi):
    """Upper-triangle relative rates (row-major i<j order) + frequencies -> unnormalised Q."""
    pi = np.asarray(pi, dtype=np.float64)
    s = pi.shape[0]
    q = np.zeros((s, s))
    k = 0
    for i in range(s):
        for j in range(i + 1, s):
            q[i, j] = rel_rates_upper[k] * pi[j]
            q[j, i] = rel_rates_upper[k] * pi[i]
            k += 1
    np.fill_diagonal(q, 0.0)
    np.fill_diagonal(q, -q.sum(axis=1))
    return q


def normalization(q, pi):
    return float(-(np.diag(q) * np.asarray(pi)).sum())


def decompose_reversible(q, pi):
    """Eigen system of a time-reversible Q, eigenvalues normalised to one expected substitution
    per unit time."""
    pi = np.asarray(pi, dtype=np.float64)
    norm = normalization(q, pi)
    sq = np.sqrt(pi)
    sym = (q * sq[:, None]) / sq[None, :]
    sym = 0.5 * (sym + sym.T)
    evals, v = np.linalg.eigh(sym)
    evec = v / sq[:, None]
    ievc = v.T * sq[None, :]
    return EigenDecomposition(evec, ievc, evals / norm)


def decompose_general(q, pi):
    """Eigen system of a (possibly non-reversible) Q with a REAL spectrum, normalised like ``decompose_reversible``
    (the reference uses this for asymmetric ``generalSubstitutionModel``s; complex spectra need EIGEN_COMPLEX, which
    this engine does not offer)."""
    pi = np.asarray(pi, dtype=np.float64)
    norm = normalization(q, pi)
    evals, evec = np.linalg.eig(np.asarray(q, dtype=np.float64))
    if np.max(np.abs(np.imag(evals))) > 1e-12:
        raise ValueError("complex eigenvalues: not supported")
    evec = np.real(evec)
    return EigenDecomposition(evec, np.linalg.inv(evec), np.real(evals) / norm)


def asymmetric_q(rates_upper_then_lower, pi):
    """S(S-1) relative rates: upper triangle row-major, then lower triangle (generalSubstitutionModel ordering);
    Q_ij = r_ij * pi_j."""
    pi = np.asarray(pi, dtype=np.float64)
    s = pi.shape[0]
    q = np.zeros((s, s))
    k = 0
    for i in range(s):
        for j in range(i + 1, s):
            q[i, j] = rates_upper_then_lower[k] * pi[j]
            k += 1
    for j in range(s):
        for i in range(j + 1, s):
            q[i, j] = rates_upper_then_lower[k] * pi[j]
            k += 1
    np.fill_diagonal(q, -q.sum(axis=1))
    return q


def gtr(rates_ac_ag_at_cg_ct_gt, pi):
    return decompose_reversible(reversible_q(rates_ac_ag_at_cg_ct_gt, pi), pi)


def hky(kappa, pi):
    return gtr([1.0, kappa, 1.0, 1.0, kappa, 1.0], pi)


def jc69():
    return hky(1.0, [0.25, 0.25, 0.25, 0.25])


def random_reversible(state_count, rng, concentration=5.0):
    """Seeded reversible model for the 20-state config (SURVEY 8d allows a seeded symmetric
    exchangeability matrix + Dirichlet frequencies in place of the WAG table)."""
    n = state_count * (state_count - 1) // 2
    rates = rng.gamma(shape=1.0, scale=1.0, size=n) + 0.05
    pi = rng.dirichlet(np.full(state_count, concentration))
    return decompose_reversible(reversible_q(rates, pi), pi), pi


# ---- GY94 codon model (61 sense codons, universal code) -------------------------------------
_BASES = "TCAG"
_AA = ("FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG")


def _codons():
    out = []
    for i, a in enumerate(_BASES):
        for j, b in enumerate(_BASES):
            for k, c in enumerate(_BASES):
                aa = _AA[16 * i + 4 * j + k]
                if aa != "*":
                    out.append((a + b + c, aa))
    return out


def gy94(kappa, omega, codon_pi=None):
    """GY94CodonModel.java:165-188: rate 0 for multi-position changes, kappa for synonymous
    transitions, 1 for synonymous transversions, kappa*omega / omega for the non-synonymous ones;
    Q_ij = rate * pi_j."""
    cods = _codons()
    s = len(cods)
    assert s == 61
    if codon_pi is None:
        codon_pi = np.full(s, 1.0 / s)
    purines = set("AG")
    rates = []
    for i in range(s):
        for j in range(i + 1, s):
            ci, ai = cods[i]
            cj, aj = cods[j]
            diff = [p for p in range(3) if ci[p] != cj[p]]
            if len(diff) != 1:
                rates.append(0.0)
                continue
            x, y = ci[diff[0]], cj[diff[0]]
            transition = (x in purines) == (y in purines)
            r = kappa if transition else 1.0
            if ai != aj:
                r *= omega
            rates.append(r)
    q = reversible_q(rates, codon_pi)
    return decompose_reversible(q, codon_pi), np.asarray(codon_pi, dtype=np.float64)


def complex_q(rates, state_count):
    """Rate matrix of dr.evomodel.substmodel.ComplexSubstitutionModel (the asymmetric discrete-trait model): S (S - 1)
    relative rates, the upper triangle in row order first, then the lower triangle COLUMN by column; no frequency scaling;
    rows sum to zero (ComplexSubstitutionModel.java:205-229 ``setupQMatrix`` with flat frequencies)."""
    s = state_count
    q = np.zeros((s, s))
    k = 0
    for i in range(s):
        for j in range(i + 1, s):
            q[i, j] = max(rates[k], 0.0); k += 1
    for j in range(s):
        for i in range(j + 1, s):
            q[i, j] = max(rates[k], 0.0); k += 1
    np.fill_diagonal(q, -q.sum(axis=1))
    return q


def stationary_distribution(q):
    """pi with pi Q = 0, sum 1 (ComplexSubstitutionModel.java:84-109 solves the same linear system)."""
    s = q.shape[0]
    a = np.vstack([q.T, np.ones(s)])
    b = np.zeros(s + 1); b[s] = 1.0
    return np.linalg.lstsq(a, b, rcond=None)[0]


def decompose_complex(q):
    """Eigen system of ANY rate matrix in the REAL BLOCK FORM BEAST hands to an EIGEN_COMPLEX BEAGLE instance
    (ComplexSubstitutionModel.java:121-173, the Colt EigenvalueDecomposition it wraps): Q V = V D with D block diagonal —
    a real eigenvalue in a 1x1 block, a conjugate pair a +/- b i as [a, b; -b, a] on two consecutive columns holding the
    real and imaginary part of the eigenvector.  Returns (pi, EigenDecomposition) with ``evals`` of length 2 S: S real parts,
    then S imaginary parts (b, -b for a pair), normalised to one expected substitution per unit time at stationarity."""
    q = np.asarray(q, dtype=np.float64)
    s = q.shape[0]
    pi = stationary_distribution(q)
    q = q / float(-(np.diag(q) * pi).sum())
    w, v = np.linalg.eig(q)
    cols, re, im = [], [], []
    used = np.zeros(s, dtype=bool)
    for k in range(s):
        if used[k]:
            continue
        used[k] = True
        if abs(w[k].imag) < 1e-14:
            cols.append(v[:, k].real); re.append(w[k].real); im.append(0.0)
            continue
        partner = [m for m in range(s) if not used[m] and abs(w[m] - np.conj(w[k])) < 1e-9][0]
        used[partner] = True
        lam, vec = (w[k], v[:, k]) if w[k].imag > 0 else (w[partner], v[:, partner])
        cols += [vec.real, vec.imag]
        re += [lam.real, lam.real]; im += [lam.imag, -lam.imag]
    vmat = np.stack(cols, axis=1)
    return pi, EigenDecomposition(vmat, np.linalg.inv(vmat), np.concatenate([re, im]))
