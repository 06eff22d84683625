"""The typed Gumbel rows of createEValueParams against Monte-Carlo on the GPU scorer.

tests/golden/gumbel_mc.json was made on an MI355X by tests/golden/make_gumbel_mc.py: for every gapped row typed into
sift4g_amd/csrc/swsharp/evalue.cpp, 6.7e7 random pairs of 350 residues and 1.7e7 pairs of 1000 residues (Robinson & Robinson
background) scored by s4g_scan under the scorer whose gap model the row describes (existence g / extension e  ==  this library's
open = g + e, extend = e), 300 pairs per row and length re-scored by the CPU oracle, the histogram of the scores, and a
maximum-likelihood fit of (lambda, K) under the finite-size corrected model evalue.cpp evaluates.  This file runs on the CPU: it
parses the rows evalue.cpp holds NOW, redoes every fit from the stored histograms, and compares.

What "inside the fit's confidence interval" can mean here: the statistical error of the fits is 0.2-0.7 % in lambda (block
spread, stored), far below the precision the published rows have -- three digits of lambda, two of K, themselves simulation
results of the 1990s.  So the bars are those of the published precision, and they are tight enough to catch a mistyped digit:
  * lambda within 1.5 % of the fit at L = 1000 (2.2 % at L = 350, where the finite-size correction carries more of the fit);
  * the chance-hit probability the typed row predicts at the score whose observed tail is 1e-4 -- the quantity `--evalue`
    thresholds -- within [0.70, 1.25] of the observed one (a K that is off by a factor 1.5, or a lambda off by 2 %, leaves it).
One table is outside those bars as PUBLISHED, not as typed: every PAM_250 row predicts MORE chance hits than occur (lambda
0.9-3.7 % low, tail ratio 1.05-1.65): conservative, consistently so over all 15 rows, and the ungapped PAM_250 row is pinned by
exact calculation (tests/test_oracle.py), so the matrix is the published one.  Its rows get their own, one-sided bars.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

import gumbel_fit as gf

EVALUE_CPP = os.path.join(ROOT, "sift4g_amd", "csrc", "swsharp", "evalue.cpp")


@pytest.fixture(scope="module")
def mc():
    return json.load(open(os.path.join(GOLDEN, "gumbel_mc.json")))


@pytest.fixture(scope="module")
def tables():
    return gf.typed_tables(EVALUE_CPP)


def _refit(entry, L, row, ungapped):
    d = entry["lengths"][str(L)]
    return gf.fit_row(np.asarray(d["hist"]), L, row, ungapped, y_offset=d["y_first"], below=[[d["below"]]])


def test_every_typed_row_was_simulated_and_scored_exactly(mc, tables):
    have = {(r["matrix"], r["existence"], r["extension"]): r for r in mc["rows"]}
    n = 0
    for matrix, rows in tables.items():
        for row in rows[1:]:
            e = have.get((matrix, row["open"], row["extension"]))
            assert e is not None, f"{matrix} {row['open']}/{row['extension']} is typed into evalue.cpp but has no Monte-Carlo entry: rerun tests/golden/make_gumbel_mc.py"
            assert (e["scorer_open"], e["scorer_extend"]) == (row["open"] + row["extension"], row["extension"])
            for L in ("350", "1000"):
                d = e["lengths"][L]
                assert d["pairs"] >= 10_000_000 and d["oracle_checked"] >= 300 and d["oracle_mismatches"] == 0
            n += 1
    assert n == len(mc["rows"]) == 82


def test_stored_fits_are_reproducible_on_the_cpu(mc, tables):
    """the (lambda, K) in the golden file are what tests/gumbel_fit.py gives from the stored histograms, here, now"""
    for e in mc["rows"][::5]:
        rows = tables[e["matrix"]]
        row = next(r for r in rows if (r["open"], r["extension"]) == (e["existence"], e["extension"]))
        for L in (350, 1000):
            f = _refit(e, L, row, rows[0])
            g = e["lengths"][str(L)]["fit"]
            if all(row[k] == e["typed"][k] for k in ("lambda", "K", "a", "Alpha", "Sigma")):     # same finite-size constants as when it was made
                assert f["lambda"] == pytest.approx(g["lambda"], rel=2e-4) and f["K"] == pytest.approx(g["K"], rel=3e-3)
                assert f["y_ref"] == g["y_ref"] and f["tail_observed"] == pytest.approx(g["tail_observed"], rel=1e-12)
            se_lam, se_lnk = gf.block_spread(g["block_fits"])
            assert se_lam == pytest.approx(g["lambda_se"], rel=1e-3) and se_lam < 0.008 * g["lambda"]      # statistical error: under 0.8 %


def test_typed_rows_agree_with_the_simulation(mc, tables):
    worst = {}
    for e in mc["rows"]:
        rows = tables[e["matrix"]]
        row = next(r for r in rows if (r["open"], r["extension"]) == (e["existence"], e["extension"]))
        pam250 = e["matrix"] == "PAM_250"
        for L, lam_tol in ((1000, 0.015), (350, 0.022)):
            f = _refit(e, L, row, rows[0])                     # with the constants evalue.cpp holds NOW
            dev = row["lambda"] / f["lambda"] - 1.0
            ratio = f["tail_ratio_typed"]
            tag = f"{e['matrix']} {e['existence']}/{e['extension']} L={L}: typed lambda {row['lambda']} K {row['K']}, fit {f['lambda']:.4f} {f['K']:.4f}, tail typed/observed {ratio:.2f}"
            if pam250:
                assert -0.04 <= dev <= 0.0, tag                 # the published rows are low (conservative), never high
                assert 1.0 <= ratio <= 1.75, tag
            else:
                assert abs(dev) <= lam_tol, tag
                assert 0.70 <= ratio <= 1.25, tag
            assert 0.93 <= f["tail_ratio_fit"] <= 1.07, tag     # the model itself describes the simulated tail
            w = worst.setdefault(e["matrix"], [0.0, 1.0, 1.0])
            w[0] = max(w[0], abs(dev)); w[1] = min(w[1], ratio); w[2] = max(w[2], ratio)
    print("\nworst |lambda typed / fit - 1| and range of tail typed / observed per table:",
          {k: (round(v[0], 4), round(v[1], 2), round(v[2], 2)) for k, v in worst.items()})


def test_a_mistyped_digit_would_be_caught(mc, tables):
    """the bars above are not vacuous: swap two digits of the default row's lambda, or take K from the next row, and they fail"""
    e = next(r for r in mc["rows"] if (r["matrix"], r["existence"], r["extension"]) == ("BLOSUM_62", 11, 1))
    rows = tables["BLOSUM_62"]
    row = dict(next(r for r in rows if (r["open"], r["extension"]) == (11, 1)))
    assert row["lambda"] == 0.267 and row["K"] == 0.041
    for bad in (dict(row, **{"lambda": 0.276}), dict(row, **{"lambda": 0.257}), dict(row, K=0.059), dict(row, K=0.024)):
        f = _refit(e, 1000, bad, rows[0])
        assert abs(bad["lambda"] / f["lambda"] - 1.0) > 0.015 or not (0.70 <= f["tail_ratio_typed"] <= 1.25), bad


def test_which_row_describes_sift4gs_default_scorer(mc, tables):
    """sift4g's defaults are BLOSUM_62 with gap open 10, extend 1 (/root/reference/sift4g/src/main.cpp:71-76), a gap of k costing
    10 + (k - 1): BLAST's existence 9 / extension 1.  The simulation of exactly that scorer reproduces the (9, 1) row -- the row the
    "blast" convention of evalue.cpp selects -- and is far from the (10, 1) row the default "swsharp" convention looks up (what
    upstream does, as far as it can be recalled, stays a switch: S4G_EVALUE_CONVENTION)."""
    e = next(r for r in mc["rows"] if r["matrix"] == "BLOSUM_62" and (r["scorer_open"], r["scorer_extend"]) == (10, 1))
    assert (e["existence"], e["extension"]) == (9, 1)
    rows = tables["BLOSUM_62"]
    r91 = next(r for r in rows if (r["open"], r["extension"]) == (9, 1))
    r101 = next(r for r in rows if (r["open"], r["extension"]) == (10, 1))
    f = e["lengths"]["1000"]["fit"]
    assert abs(r91["lambda"] / f["lambda"] - 1.0) < 0.01 and abs(np.log(r91["K"] / f["K"])) < 0.15
    assert r101["lambda"] / f["lambda"] - 1.0 > 0.15
    # what the lookup convention costs at the default threshold: chance hits at the score where the (10, 1) row says E = 1e-4
    c = gf.fsc_constants(r91, rows[0])
    c10 = gf.fsc_constants(r101, rows[0])
    y = np.arange(40, 400)
    E10 = gf.area(y, 350, 340, c10) * r101["K"] * np.exp(-r101["lambda"] * y) * (34_548_336 / 340.0)
    y_cut = int(y[np.argmax(E10 <= 1e-4)])
    E9 = float(gf.area(y_cut, 350, 340, c) * r91["K"] * np.exp(-r91["lambda"] * y_cut) * (34_548_336 / 340.0))
    assert 3.0 < E9 / 1e-4 < 300.0          # the swsharp-convention cut lets through hits whose true e-value is 1-2 orders larger
