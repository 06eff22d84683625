"""Independent restatement of the e-value statistics behind sift4g's `--evalue` cut -- TEST INFRASTRUCTURE ONLY.

Path it checks: createEValueParams / eValues of the swsharp-compatible library (sift4g_amd/csrc/swsharp/evalue.cpp), called by
the reference at /root/reference/sift4g/src/main.cpp:207-210 and database_alignment.cpp:129-134.  swsharp's own evalue source
is an absent submodule, so this is NOT a restatement of reference code: it is written from the published mathematics,

  * Karlin & Altschul 1990 (PNAS 87:2264): ungapped lambda, relative entropy H and K of a scoring system (lattice case);
  * Park, Sheetlin, Ma, Madden, Spouge 2012 (BMC Res Notes 5:286): finite-size correction -- the numbers of residues the
    optimal alignment leaves out of each sequence are approximately bivariate normal with means a y + b, variances
    alpha y + beta and covariance sigma y + tau, and E = K exp(-lambda y) E[(m - L_I)^+ (n - L_J)^+],

in a different form from evalue.cpp (expected positive part of a normal variable, scipy-free closed forms, no shared code, no
shared constants file): the two agree to ~1e-13 relative or the test fails.  "Parity unpinned" all the same: the constants of
the gapped rows are published simulation results typed in from memory of the NCBI tables; only the ungapped lambda / H / K
(recomputed here from the matrices themselves) are pinned by calculation.

Importers allowed: tests/ and tests/golden/make_evalue_kat.py.
"""
from math import erfc, exp, log, pi, sqrt

import numpy as np

# Robinson & Robinson 1991 amino-acid background, order ARNDCQEGHILKMFPSTWYV (the background NCBI's tables were computed with)
ROBINSON = np.array([0.07805, 0.05129, 0.04487, 0.05364, 0.01925, 0.04264, 0.06295, 0.07377, 0.02199, 0.05142,
                     0.09019, 0.05744, 0.02243, 0.03856, 0.05203, 0.07120, 0.05841, 0.01330, 0.03216, 0.06441])
AA20 = "ARNDCQEGHILKMFPSTWYV"

# (matrix, open, extension) -> (lambda, K, H, alpha, alpha_v, sigma); (matrix, 0, 0) = ungapped.  Typed from the published NCBI
# table for the rows sift4g's defaults and the tests touch; deliberately a second, separate typing from evalue.cpp's.
PUBLISHED = {
    ("BLOSUM_62", 0, 0): (0.3176, 0.134, 0.4012, 0.7916, 4.964660, 4.964660),
    ("BLOSUM_62", 11, 1): (0.267, 0.041, 0.14, 1.9, 42.602800, 43.636200),
    ("BLOSUM_62", 10, 1): (0.243, 0.024, 0.10, 2.5, 83.178700, 84.065100),
    ("BLOSUM_62", 9, 1): (0.206, 0.010, 0.052, 4.0, 210.333000, 214.842000),
    ("BLOSUM_62", 11, 2): (0.297, 0.082, 0.27, 1.1, 12.673800, 12.757600),
    ("BLOSUM_62", 10, 2): (0.291, 0.075, 0.23, 1.3, 16.474000, 16.602600),
    ("BLOSUM_45", 0, 0): (0.2291, 0.0924, 0.2514, 0.9113, 9.611060, 9.611060),
    ("BLOSUM_45", 14, 2): (0.195, 0.032, 0.10, 1.9, 60.736200, 61.102300),
    ("BLOSUM_50", 0, 0): (0.2318, 0.112, 0.3362, 0.6895, 5.388310, 5.388310),
    ("BLOSUM_50", 13, 2): (0.193, 0.035, 0.12, 1.6, 42.059500, 42.471600),
    ("BLOSUM_80", 0, 0): (0.3430, 0.177, 0.6568, 0.5222, 1.918130, 1.918130),
    ("BLOSUM_80", 10, 1): (0.299, 0.071, 0.27, 1.1, 13.833800, 14.533400),
    ("BLOSUM_90", 0, 0): (0.3346, 0.190, 0.7547, 0.4434, 1.377760, 1.377760),
    ("BLOSUM_90", 10, 1): (0.290, 0.075, 0.28, 1.04, 11.483900, 12.269800),
    ("PAM_30", 0, 0): (0.3400, 0.283, 1.754, 0.1938, 0.161818, 0.161818),
    ("PAM_30", 9, 1): (0.294, 0.11, 0.61, 0.48, 2.463920, 3.186150),
    ("PAM_70", 0, 0): (0.3345, 0.229, 1.029, 0.3250, 0.633439, 0.633439),
    ("PAM_70", 10, 1): (0.291, 0.091, 0.41, 0.71, 6.002970, 6.716570),
    ("PAM_250", 0, 0): (0.2252, 0.0868, 0.2223, 0.98, 11.754300, 11.754300),
    ("PAM_250", 14, 2): (0.182, 0.024, 0.073, 2.5, 96.315100, 96.756800),
}


# ------------------------------------------------------------------------------------------------------------------
# Karlin-Altschul statistics of an ungapped scoring system, from the matrix itself

def score_distribution(table26, freqs=ROBINSON):
    """P(score = s) for two residues drawn independently from `freqs` -> (lowest score, probabilities[lowest..highest])"""
    ix = [ord(c) - 65 for c in AA20]
    S = np.asarray(table26)[np.ix_(ix, ix)].astype(int)
    p = np.asarray(freqs, dtype=float)
    p = p / p.sum()
    lo, hi = int(S.min()), int(S.max())
    prob = np.zeros(hi - lo + 1)
    np.add.at(prob, (S - lo).ravel(), np.outer(p, p).ravel())
    return lo, prob


def ungapped_lambda(lo, prob):
    """the unique positive root of sum_s P(s) exp(lambda s) = 1 (bisection; the function is convex and dips below 1 first)"""
    s = np.arange(lo, lo + len(prob), dtype=float)
    assert (prob * s).sum() < 0 and s.max() > 0, "expected score must be negative and a positive score possible"
    f = lambda l: float((prob * np.exp(l * s)).sum()) - 1.0
    a, b = 1e-6, 1.0
    while f(b) < 0:
        b *= 2.0
    while f(a) > 0:
        a /= 2.0
    for _ in range(200):
        mid = 0.5 * (a + b)
        if f(mid) < 0:
            a = mid
        else:
            b = mid
    return 0.5 * (a + b)


def ungapped_entropy(lo, prob, lam):
    """H = lambda * sum_s s P(s) exp(lambda s)   (nats per aligned pair)"""
    s = np.arange(lo, lo + len(prob), dtype=float)
    return lam * float((prob * s * np.exp(lam * s)).sum())


def ungapped_K(lo, prob, lam, H, tol=1e-12, max_terms=400):
    """Karlin & Altschul's K for a lattice of span 1:
         K = exp(-2 sum_{k>=1} (1/k) (E[exp(lambda S_k); S_k < 0] + P(S_k >= 0))) / ((H / lambda) (1 - exp(-lambda)))
       with S_k the sum of k independent pair scores (distribution by repeated convolution)."""
    pk = np.array([1.0])
    low_k = 0
    outer = 0.0
    for k in range(1, max_terms + 1):
        pk = np.convolve(pk, prob)
        low_k += lo
        s = np.arange(low_k, low_k + len(pk), dtype=float)
        neg = s < 0
        inner = float((pk[neg] * np.exp(lam * s[neg])).sum() + pk[~neg].sum())
        outer += inner / k
        if inner < tol:
            break
    return exp(-2.0 * outer) / ((H / lam) * (1.0 - exp(-lam)))


def ungapped_statistics(table26, freqs=ROBINSON):
    lo, prob = score_distribution(table26, freqs)
    lam = ungapped_lambda(lo, prob)
    H = ungapped_entropy(lo, prob, lam)
    return dict(lam=lam, H=H, K=ungapped_K(lo, prob, lam, H))


# ------------------------------------------------------------------------------------------------------------------
# Finite-size corrected e-value

def _Phi(z):
    return 0.5 * erfc(-z / sqrt(2.0))


def _positive_part(limit, mean, var):
    """E[(limit - X)^+] for X ~ Normal(mean, var)"""
    sd = sqrt(var)
    z = (limit - mean) / sd
    return (limit - mean) * _Phi(z) + sd * exp(-0.5 * z * z) / sqrt(2.0 * pi)


def gumbel_parameters(matrix, gap_open, gap_extend, convention="swsharp"):
    """-> dict(lam, K, a, b, alpha, beta, sigma, tau) for a scorer with cost(gap of k) = gap_open + (k-1) gap_extend.
    convention "swsharp": the published row whose numbers equal (gap_open, gap_extend); "blast": the row for existence
    = gap_open - gap_extend (see evalue.cpp for why both exist).  KeyError if this module holds no such row."""
    first = gap_open if convention == "swsharp" else gap_open - gap_extend
    lam, K, H, a, alpha, sigma = PUBLISHED[(matrix, first, gap_extend)]
    _, _, _, a_un, alpha_un, _ = PUBLISHED[(matrix, 0, 0)]
    G = first + gap_extend
    return dict(lam=lam, K=K, a=a, b=2.0 * G * (a_un - a), alpha=alpha, beta=2.0 * G * (alpha_un - alpha),
                sigma=sigma, tau=2.0 * G * (alpha_un - sigma))


def evalue(score, m, n, db_residues, P):
    """Expected number of chance alignments scoring >= score between a query of m residues and a database of db_residues,
    evaluated on a target of n residues and scaled to the database (what sift4g compares with --evalue)."""
    y = float(score)
    mean = P["a"] * y + P["b"]                                        # symmetric matrix: the same law for both sequences
    var = max(2.0 * P["alpha"] / P["lam"], P["alpha"] * y + P["beta"])
    cov = max(2.0 * P["sigma"] / P["lam"], P["sigma"] * y + P["tau"])
    sd = sqrt(var)
    area = (_positive_part(m, mean, var) * _positive_part(n, mean, var)
            + cov * _Phi((m - mean) / sd) * _Phi((n - mean) / sd))
    return area * P["K"] * exp(-P["lam"] * y) * (float(db_residues) / float(max(n, 1)))


def bit_score(score, P):
    return (P["lam"] * score - log(P["K"])) / log(2.0)
