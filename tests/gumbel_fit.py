"""Fit of (lambda, K) to simulated optimal local alignment scores -- shared by tests/golden/make_gumbel_mc.py (makes
tests/golden/gumbel_mc.json on the GPU) and tests/test_gumbel_mc.py (redoes the fit from the stored histograms on the CPU and
compares it with the rows typed into sift4g_amd/csrc/swsharp/evalue.cpp).  numpy + scipy only.

Model (Altschul & Gish 1996; Park, Sheetlin, Ma, Madden, Spouge 2012): for two random sequences of m and n residues the number
of distinct local alignments scoring >= y is Poisson with mean

    E(y) = area(m, n, y) K exp(-lambda y),     P(S < y) = exp(-E(y))

where area is the finite-size corrected search space -- E[(m - L_I)^+ (n - L_J)^+] for bivariate-normal alignment extents with
mean a y + b, variance alpha y + beta, covariance sigma y + tau -- exactly as evalue.cpp evaluates it (same expressions, written
for numpy arrays).  a, alpha, sigma come from the typed row, b / beta / tau from the row and the matrix's ungapped row.
"""
import re

import numpy as np
from scipy.optimize import minimize
from scipy.special import erfc

FIELDS = ("open", "extension", "lambda", "K", "H", "a", "Alpha", "Sigma")


def typed_tables(evalue_cpp):
    """{canonical matrix name: [row dict, ...]} of every GumbelRow typed into evalue.cpp; row 0 = the ungapped one."""
    src = open(evalue_cpp).read()
    arrays = {}
    for m in re.finditer(r"const GumbelRow (k\w+)\[\] = \{(.*?)\n\};", src, re.S):
        rows = []
        for r in re.finditer(r"\{([^{}]+)\}", m.group(2)):
            v = [float(x) for x in r.group(1).split(",")]
            assert len(v) == 8, r.group(0)
            d = dict(zip(FIELDS, v))
            d["open"], d["extension"] = int(d["open"]), int(d["extension"])
            rows.append(d)
        assert rows and rows[0]["open"] == 0 and rows[0]["extension"] == 0, m.group(1)
        arrays[m.group(1)] = rows
    tables = {}
    for m in re.finditer(r'\{"(\w+)", S4G_ROWS\((k\w+)\), (\d+)\}', src):
        tables[m.group(1)] = arrays[m.group(2)]
    assert tables, "no GumbelTable found in " + evalue_cpp
    return tables


def fsc_constants(row, ungapped):
    """b, beta, tau of createEValueParams (evalue.cpp): G = existence + extension of the row."""
    Gp = float(row["open"] + row["extension"])
    return dict(a=row["a"], b=2.0 * Gp * (ungapped["a"] - row["a"]), alpha=row["Alpha"], beta=2.0 * Gp * (ungapped["Alpha"] - row["Alpha"]),
                sigma=row["Sigma"], tau=2.0 * Gp * (ungapped["Alpha"] - row["Sigma"]), lam_floor=row["lambda"])


def area(y, m, n, c):
    """Spouge's finite-size corrected search space for score(s) y (array) and sequence lengths m, n."""
    y = np.asarray(y, dtype=float)
    shift = c["a"] * y + c["b"]
    v = np.maximum(c["alpha"] * y + c["beta"], 2.0 * c["alpha"] / c["lam_floor"])
    sv = np.sqrt(v)
    inv = 0.39894228040143267793994605993438

    def part(length):
        li = length - shift
        z = li / sv
        P = 0.5 * erfc(-z / np.sqrt(2.0))
        return li * P + sv * inv * np.exp(-0.5 * z * z), P
    p1, Pm = part(m)
    p2, Pn = part(n)
    cy = np.maximum(c["sigma"] * y + c["tau"], 2.0 * c["sigma"] / c["lam_floor"])
    return p1 * p2 + cy * Pm * Pn


def _tail_start(total, frac):
    tail = np.cumsum(total[::-1])[::-1] / total.sum()
    return int(np.argmax(tail <= frac))


def mle(counts_tail, n_below, y0, L, c, start):
    """Maximum-likelihood (lambda, ln K) from counts_tail[k] = pairs with S = y0 + k (the last bin: S >= its score) and n_below pairs
    with S < y0."""
    ys = np.arange(y0, y0 + len(counts_tail) + 1, dtype=float)          # bin edges y0 .. y0 + n
    ar = area(ys[:-1], L, L, c)
    ar_edges = np.append(ar, 0.0)
    cnt = np.asarray(counts_tail, dtype=float)

    def nll(theta):
        lam, lnK = theta
        E = ar_edges * np.exp(lnK - lam * ys)                             # E at every lower edge; 0 beyond the last bin
        F = np.exp(-E)                                                    # P(S < edge); F beyond the last = 1
        F[-1] = 1.0
        p = np.maximum(F[1:] - F[:-1], 1e-300)
        return -(n_below * (-E[0]) + float((cnt * np.log(p)).sum()))
    r = minimize(nll, np.array([start[0], np.log(start[1])]), method="Nelder-Mead", options=dict(xatol=1e-7, fatol=1e-7, maxiter=4000))
    return float(r.x[0]), float(r.x[1])


def block_spread(per):
    """per[G][G] = (lambda, ln K) fitted on block (i, j) alone.  Blocks on one diagonal (i, (i + s) mod G) share no sequence:
    the standard error of a diagonal's mean, averaged over the G diagonals = the error of 1/G of the pairs (conservative)."""
    per = np.asarray(per, dtype=float)
    G = per.shape[0]
    se = np.zeros((G, 2))
    for s in range(G):
        d = np.array([per[i, (i + s) % G] for i in range(G)])
        se[s] = d.std(axis=0, ddof=1) / np.sqrt(G)
    return [float(x) for x in se.mean(axis=0)]


def fit_row(hist, L, row, ungapped, tail_frac=0.02, y_offset=0, below=None):
    """hist[G][G][score - y_offset] = pairs of block (i, j) with that score (`below`[G][G] = pairs under y_offset, if the histogram
    was cut); or hist[score - y_offset] over all pairs (then no block fits).  -> the fit on all pairs, its block-spread standard
    errors, and how the TYPED row predicts the observed tail."""
    hist = np.asarray(hist, dtype=np.int64)
    blocks = hist.ndim == 3
    if not blocks:
        hist = hist[None, None, :]
    G = hist.shape[0]
    below = np.zeros((G, G), dtype=np.int64) if below is None else np.asarray(below, dtype=np.int64).reshape(G, G)
    c = fsc_constants(row, ungapped)
    total = hist.sum(axis=(0, 1))
    n_pairs = int(total.sum() + below.sum())
    tail = (np.cumsum(total[::-1])[::-1]) / float(n_pairs)
    k0 = int(np.argmax(tail <= tail_frac))
    y0 = y_offset + k0
    # the last bin collects everything from the first score whose count falls under 5
    small = np.nonzero(total[k0:] < 5)[0]
    k1 = k0 + (int(small[0]) if len(small) else len(total) - k0)
    k1 = max(k1, k0 + 4)

    def pack(h, b):
        t = h[k0:k1].astype(float).copy()
        t[-1] += h[k1:].sum()
        return t, float(b + h[:k0].sum())
    ct, nb = pack(total, below.sum())
    lam, lnK = mle(ct, nb, y0, L, c, (row["lambda"], row["K"]))
    per, lam_se, lnK_se = None, None, None
    if blocks:
        per = np.zeros((G, G, 2))
        for i in range(G):
            for j in range(G):
                t, b = pack(hist[i, j], below[i, j])
                per[i, j] = mle(t, b, y0, L, c, (lam, np.exp(lnK)))
        lam_se, lnK_se = block_spread(per)
    # operational check: the tail probability the typed row predicts at the score whose observed tail is nearest 1e-4
    k_ref = int(np.argmin(np.abs(np.log(np.maximum(tail, 1e-300)) - np.log(1e-4))))
    y_ref = y_offset + k_ref
    obs = float(tail[k_ref])
    E_typed = float(area(y_ref, L, L, c) * row["K"] * np.exp(-row["lambda"] * y_ref))
    E_fit = float(area(y_ref, L, L, c) * np.exp(lnK - lam * y_ref))
    return {"lambda": lam, "K": float(np.exp(lnK)), "lambda_se": lam_se, "lnK_se": lnK_se,
            "block_fits": None if per is None else np.round(per, 5).tolist(), "y0": y0, "y_last": y_offset + k1 - 1,
            "tail_pairs": int(ct.sum()), "pairs": n_pairs, "y_ref": y_ref, "tail_observed": obs,
            "tail_typed": 1.0 - float(np.exp(-E_typed)), "tail_fit": 1.0 - float(np.exp(-E_fit)),
            "tail_ratio_typed": (1.0 - float(np.exp(-E_typed))) / obs, "tail_ratio_fit": (1.0 - float(np.exp(-E_fit))) / obs}
