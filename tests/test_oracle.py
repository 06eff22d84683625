"""CPU tests that pin the oracle (oracle/sw_oracle.c).

The reference holds no golden vectors for this path: the oracle is "parity unpinned"
against swsharp and is pinned instead by hand-computed known answers, the published
Karlin-Altschul statistics of BLOSUM62, and identities on the reference's own input fixtures.
"""
import json
import os
import sys

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import ROOT, rand_seqs

sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def test_micro_known_answers(oracle_mod, blosum62, micro):
    for c in micro["cases"]:
        got = oracle_mod.sw_score(oracle_mod.encode(c["q"]), oracle_mod.encode(c["t"]), blosum62,
                                  micro["gap_open"], micro["gap_extend"])
        assert got == c["score"], (c["name"], got, c["score"])


def test_blosum62_shape_and_symmetry(blosum62):
    assert blosum62.shape == (26, 26)
    assert (blosum62 == blosum62.T).all()
    idx = {c: ord(c) - 65 for c in "ABCDEFGHIJKLMNOPQRSTUVWXYZ"}
    assert blosum62[idx["W"], idx["W"]] == 11 and blosum62[idx["C"], idx["C"]] == 9
    assert blosum62[idx["A"], idx["R"]] == -1 and blosum62[idx["L"], idx["S"]] == -2
    # J, O, U have no NCBI row: documented assumption = X row
    for c in "JOU":
        assert (blosum62[idx[c]] == blosum62[idx["X"]]).all()


def test_blosum62_published_statistics(blosum62):
    """Ungapped Karlin-Altschul lambda and relative entropy of BLOSUM62 under the Robinson & Robinson
    background: published values lambda = 0.3176, H = 0.4012 (NCBI BLAST)."""
    from scipy.optimize import brentq
    aa = "ARNDCQEGHILKMFPSTWYV"
    p = np.array([0.07805, 0.05129, 0.04487, 0.05364, 0.01925, 0.04264, 0.06295, 0.07377, 0.02199, 0.05142,
                  0.09019, 0.05744, 0.02243, 0.03856, 0.05203, 0.07120, 0.05841, 0.01330, 0.03216, 0.06441])
    p = p / p.sum()
    ix = [ord(c) - 65 for c in aa]
    S = blosum62[np.ix_(ix, ix)].astype(float)
    pp = np.outer(p, p)
    lam = brentq(lambda l: (pp * np.exp(l * S)).sum() - 1.0, 0.05, 1.0)
    H = lam * (pp * np.exp(lam * S) * S).sum()
    assert abs(lam - 0.3176) < 5e-4
    assert abs(H - 0.4012) < 1e-3


MATRIX_NAMES = ["BLOSUM_45", "BLOSUM_50", "BLOSUM_62", "BLOSUM_80", "BLOSUM_90", "PAM_30", "PAM_70", "PAM_250"]


def test_published_statistics_of_every_matrix():
    """Every built-in table (the five BLOSUMs and three PAMs the CLI offers, main.cpp:304-314) is pinned by the ungapped
    Karlin-Altschul statistics NCBI publishes for it: lambda, relative entropy H and K, all three recomputed from the typed-in
    table itself by oracle/evalue_ref.py (lambda and H to four digits, K to the three digits that are published).  This fixes
    the 20 x 20 core of each table; and the published a = lambda / H of the ungapped rows (PAM_250's published 0.98 is the one
    that does not follow that rule)."""
    from oracle import evalue_ref as er
    from sift4g_amd.device import matrix_by_name
    for name in MATRIX_NAMES:
        M = matrix_by_name(name)
        assert (M == M.T).all(), name
        st_ = er.ungapped_statistics(M)
        lam, K, H, a, _, _ = er.PUBLISHED[(name, 0, 0)]
        assert abs(st_["lam"] - lam) < 1.5e-4, (name, st_)
        assert abs(st_["H"] - H) < 1e-3 * max(1.0, H), (name, st_)
        assert abs(st_["K"] - K) < 6e-4 * max(1.0, K / 0.1), (name, st_)
        if name != "PAM_250":
            assert abs(st_["lam"] / st_["H"] - a) < 3e-4, (name, st_)
    # the CLI's own spelling of the PAM tables (main.cpp:310-312) names the same tables
    for alias, name in (("BLOSUM_30", "PAM_30"), ("BLOSUM_70", "PAM_70"), ("BLOSUM_250", "PAM_250")):
        assert (matrix_by_name(alias) == matrix_by_name(name)).all()


def test_matrix_digests_are_stable():
    """SHA-256 of every table in the canonical form of tools/matrix_digest.py (what to compare with NCBI's files, which this
    image does not hold): a change to any typed-in entry shows up here."""
    import json, os, sys
    from conftest import GOLDEN, ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import matrix_digest as md
    from sift4g_amd.device import matrix_by_name
    want = json.load(open(os.path.join(GOLDEN, "matrix_sha256.json")))
    assert sorted(want) == sorted(MATRIX_NAMES)
    for name in MATRIX_NAMES:
        assert md.digest(md.canonical_from_table26(matrix_by_name(name))) == want[name], name
    # the file-format reader reduces an NCBI-style rendering of the same table to the same digest
    M = matrix_by_name("BLOSUM_62")
    syms = md.ORDER[:-1]
    text = "# Matrix made by matblas from blosum62.iij\n   " + "  ".join(syms) + "  *\n" + "".join(
        a + " " + " ".join(f"{int(M[ord(a) - 65][ord(b) - 65]):2d}" for b in syms) + " -4\n" for a in syms) + "* " + " -4" * 23 + "  1\n"
    assert md.digest(md.canonical_from_ncbi_file(text)) == want["BLOSUM_62"]


def test_three_scorers_triangulate(oracle_mod):
    """oracle/sw_oracle.c (row-major scalar), oracle/sw_simd.c (column-major AVX2 int16) and oracle/sw_numpy.py (anti-diagonal
    numpy int64): three forms of the same recurrence written separately must agree on random, homologous and degenerate
    pairs, for every built-in matrix and several gap models."""
    from oracle import sw_numpy
    from sift4g_amd.device import matrix_by_name
    rng = np.random.default_rng(77)
    qs = rand_seqs(rng, 5, 1, 140, 26) + [np.zeros(0, dtype=np.uint8), np.full(37, 22, dtype=np.uint8)]
    ts = rand_seqs(rng, 14, 1, 170, 26) + [np.zeros(0, dtype=np.uint8)]
    ts[0] = qs[0].copy()
    ts[1] = np.concatenate([qs[1][:40], rand_seqs(rng, 1, 7, 7)[0], qs[1][40:]])       # insertion
    ts[2] = np.concatenate([qs[2][:30], qs[2][41:]]) if len(qs[2]) > 50 else ts[2]    # deletion
    for name, go, ge in (("BLOSUM_62", 10, 1), ("BLOSUM_45", 14, 2), ("BLOSUM_80", 10, 1), ("PAM_30", 9, 1), ("PAM_250", 14, 2),
                         ("BLOSUM_50", 3, 3), ("BLOSUM_90", 12, 0), ("PAM_70", 1, 1)):
        M = matrix_by_name(name)
        a = oracle_mod.scan(qs, ts, M, go, ge, threads=4)
        b = sw_numpy.scan(qs, ts, M, go, ge)
        assert np.array_equal(a, b), (name, go, ge, np.argwhere(a != b)[:4].tolist())
        if oracle_mod.simd_available():
            qc, qo = oracle_mod._flat(qs)
            tc, to = oracle_mod._flat(ts)
            c = oracle_mod.scan_simd_flat(qc, qo, tc, to, M, go, ge, threads=2)
            assert np.array_equal(a, c), (name, go, ge)
    assert a.max() > 0


def test_cfg1_fixture_identities(oracle_mod, blosum62, cfg1):
    """DB entry QUERY == query LACI_ECOLI; DB LACI_ECOLI differs at 286 (L->S);
    DB PURR_SALTY == query PURR_SALTY."""
    q = {n.split()[0]: c for n, c in cfg1["queries"]}
    t = {n.split()[0]: c for n, c in cfg1["targets"]}
    M = blosum62
    self_laci = int(sum(M[a, a] for a in q["LACI_ECOLI"]))
    self_purr = int(sum(M[a, a] for a in q["PURR_SALTY"]))
    assert np.array_equal(t["QUERY"], q["LACI_ECOLI"])
    assert np.array_equal(t["PURR_SALTY"], q["PURR_SALTY"])
    diff = np.nonzero(t["LACI_ECOLI"] != q["LACI_ECOLI"])[0]
    assert diff.tolist() == [285]
    L, S = ord("L") - 65, ord("S") - 65
    assert q["LACI_ECOLI"][285] == L and t["LACI_ECOLI"][285] == S
    assert oracle_mod.sw_score(q["LACI_ECOLI"], t["QUERY"], M) == self_laci
    assert oracle_mod.sw_score(q["PURR_SALTY"], t["PURR_SALTY"], M) == self_purr
    assert oracle_mod.sw_score(q["LACI_ECOLI"], t["LACI_ECOLI"], M) == self_laci - M[L, L] + M[L, S]


def test_cfg1_golden_scores(oracle_mod, blosum62, cfg1):
    got = oracle_mod.scan([c for _, c in cfg1["queries"]], [c for _, c in cfg1["targets"]], blosum62, 10, 1, threads=2)
    assert np.array_equal(got, cfg1["scores"])
    assert cfg1["gold"]["query_lengths"] == [360, 341]
    assert sum(cfg1["gold"]["target_lengths"]) == 3091


def _rescore(path, q, t, qs, ts, M, go, ge):
    """Score an alignment path independently of the DP."""
    i, j, s, prev = qs, ts, 0, 0
    for mv in path:
        if mv == 0:
            s += int(M[q[i], t[j]]); i += 1; j += 1
        elif mv == 1:
            s -= ge if prev == 1 else go; j += 1
        else:
            s -= ge if prev == 2 else go; i += 1
        prev = mv
    return s, i - 1, j - 1


def test_traceback_rescoring(oracle_mod, blosum62):
    rng = np.random.default_rng(7)
    for _ in range(60):
        q = rand_seqs(rng, 1, 1, 80)[0]
        t = q.copy() if rng.random() < 0.3 else rand_seqs(rng, 1, 1, 90)[0]
        if rng.random() < 0.5 and len(q) > 10:   # plant an indel so gaps are exercised
            cut = int(rng.integers(3, len(q) - 3))
            t = np.concatenate([q[:cut], rand_seqs(rng, 1, 1, 4)[0], q[cut:]])
        a = oracle_mod.sw_align(q, t, blosum62, 10, 1)
        assert a["score"] == oracle_mod.sw_score(q, t, blosum62, 10, 1)
        if a["score"] == 0:
            assert len(a["path"]) == 0
            continue
        s, qe, te = _rescore(a["path"], q, t, a["q_start"], a["t_start"], blosum62, 10, 1)
        assert (s, qe, te) == (a["score"], a["q_end"], a["t_end"])
        assert a["path"][0] == 0 and a["path"][-1] == 0     # local alignments start and end on a match column


codes = st.lists(st.integers(0, 25), min_size=0, max_size=40).map(lambda l: np.array(l, dtype=np.uint8))


@settings(max_examples=150, deadline=None)
@given(q=codes, t=codes, go=st.integers(1, 14), ge=st.integers(0, 3))
def test_properties(oracle_mod, blosum62, q, t, go, ge):
    ge = min(ge, go)
    s = oracle_mod.sw_score(q, t, blosum62, go, ge)
    assert s >= 0
    assert s == oracle_mod.sw_score(t, q, blosum62, go, ge)          # symmetric matrix -> symmetric score
    assert s == oracle_mod.sw_score(q[::-1], t[::-1], blosum62, go, ge)  # reversal invariance
    if len(q):
        assert oracle_mod.sw_score(q, q, blosum62, go, ge) >= max(0, int(sum(blosum62[a, a] for a in q)))
    # a cheaper gap model can only raise the score
    assert oracle_mod.sw_score(q, t, blosum62, max(go - 1, ge), ge) >= s
    # appending residues never lowers a local score
    assert oracle_mod.sw_score(np.concatenate([q, t[:3]]), t, blosum62, go, ge) >= s


def test_pairs_and_scan_agree(oracle_mod, blosum62):
    rng = np.random.default_rng(3)
    qs, ts = rand_seqs(rng, 5, 0, 70, 26), rand_seqs(rng, 9, 0, 90, 26)
    full = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=3)
    for i, q in enumerate(qs):
        for j, t in enumerate(ts):
            assert full[i, j] == oracle_mod.sw_score(q, t, blosum62, 10, 1)
    from oracle import _flat
    qc, qo = _flat(qs); tc, to = _flat(ts)
    pq = np.array([4, 0, 2], dtype=np.uint32); pt = np.array([8, 3, 3], dtype=np.uint32)
    got = oracle_mod.pairs_flat(qc, qo, tc, to, pq, pt, blosum62, 10, 1, threads=2)
    assert got.tolist() == [full[4, 8], full[0, 3], full[2, 3]]


def test_simd_port_equals_scalar_oracle(oracle_mod, blosum62):
    """oracle/sw_simd.c (the timed CPU baseline) against the scalar restatement: every letter code, empty and ragged
    inputs, target counts that are not a multiple of the vector width, and scores beyond int16 (scalar recheck)."""
    if not oracle_mod.simd_available():
        pytest.skip("no AVX2 on this CPU")
    from oracle import _flat
    rng = np.random.default_rng(51)
    W = ord("W") - 65
    qs = rand_seqs(rng, 7, 0, 300, 26) + [np.full(3100, W, dtype=np.uint8), np.zeros(0, dtype=np.uint8)]
    ts = rand_seqs(rng, 83, 0, 400, 26) + [np.full(3300, W, dtype=np.uint8), np.zeros(0, dtype=np.uint8), qs[0][5:].copy()]
    qc, qo = _flat(qs); tc, to = _flat(ts)
    for go, ge in ((10, 1), (5, 5), (14, 0)):
        a = oracle_mod.scan_flat(qc, qo, tc, to, blosum62, go, ge, threads=4)
        b = oracle_mod.scan_simd_flat(qc, qo, tc, to, blosum62, go, ge, threads=3)          # the ladder: 8-bit -> int16 -> scalar
        assert np.array_equal(a, b), (go, ge, np.argwhere(a != b)[:5])
        c = oracle_mod.scan_simd16_flat(qc, qo, tc, to, blosum62, go, ge, threads=3)        # the int16 tier alone
        assert np.array_equal(a, c), (go, ge, np.argwhere(a != c)[:5])
    assert a[7, 83] == 3100 * 11          # > 32767: went through the scalar recheck


def test_simd_8bit_tier_and_its_escalation(oracle_mod, blosum62):
    """the AVX2 port's 8-bit tier (32 unsigned lanes, scores biased by -min score, the swimd / OPAL scheme).
    A lane is exact while its maximum stays below 255 - bias and is recomputed in int16 otherwise: poly-W runs whose scores sit
    on both sides of that mark (BLOSUM62: bias 4, mark 251 = 22.8 W), planted homologs far beyond it, a target count that is not
    a multiple of 32, every built-in matrix (PAM_30: bias 17), and penalties a byte cannot hold (the tier steps aside)."""
    if not oracle_mod.simd_available():
        pytest.skip("no AVX2 on this CPU")
    from oracle import _flat
    from sift4g_amd.device import matrix_by_name
    rng = np.random.default_rng(52)
    W = ord("W") - 65
    qs = rand_seqs(rng, 6, 1, 260, 26) + [np.full(n, W, dtype=np.uint8) for n in (22, 23, 24, 60)]
    ts = rand_seqs(rng, 77, 0, 330, 26) + [np.full(n, W, dtype=np.uint8) for n in (22, 23, 40)] + [qs[0].copy(), qs[1][3:].copy()]
    qc, qo = _flat(qs); tc, to = _flat(ts)
    exp = oracle_mod.scan_flat(qc, qo, tc, to, blosum62, 10, 1, threads=4)
    got, esc = oracle_mod.scan_simd_ladder_flat(qc, qo, tc, to, blosum62, 10, 1, threads=3)
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:5]
    assert exp[6, 77] == 242 and exp[7, 78] == 253 and exp[8, 79] == 264          # 22 / 23 / 24 W: below, above, above the mark
    assert esc == int((exp >= 251).sum()) and esc >= 5                              # exactly the pairs at or beyond the mark escalate
    for name, go, ge in (("BLOSUM_45", 14, 2), ("BLOSUM_80", 10, 1), ("PAM_30", 9, 1), ("PAM_250", 14, 2), ("BLOSUM_50", 3, 3),
                         ("BLOSUM_90", 12, 0), ("PAM_70", 1, 1), ("BLOSUM_62", 127, 127), ("BLOSUM_62", 128, 1), ("BLOSUM_62", 300, 200)):
        M = matrix_by_name(name)
        exp = oracle_mod.scan_flat(qc, qo, tc, to, M, go, ge, threads=4)
        got, esc = oracle_mod.scan_simd_ladder_flat(qc, qo, tc, to, M, go, ge, threads=3)
        assert np.array_equal(got, exp), (name, go, ge, np.argwhere(got != exp)[:5])
        if go > 127:
            assert esc == 0                                                          # the 8-bit tier stepped aside: int16 for everything
        else:
            assert esc == int((exp >= 255 + int(M.min())).sum()), (name, go, ge)


# ---- the k-mer prefilter: oracle/kmer_prefilter.c against what the reference's own code printed ----

def _prefilter_case(c):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_prefilter_golden as g
    queries, targets = g.make_case(c["seed"], c["nq"], c["nt"], c["repeats"])
    return [oracle.encode(q) for q in queries], [oracle.encode(t) for t in targets]


def test_prefilter_matches_the_reference_code():
    """tests/golden/prefilter.json holds the candidate lists the reference's database_search.cpp + hash.cpp returned (at 1 and
    4 threads) on seeded inputs (tests/golden/make_prefilter_golden.py).  Where max_candidates did not bind the restatement
    must return exactly that list; where it did, the same number of candidates, every one the reference keeps regardless of
    its thread count, and nothing below the cut."""
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "prefilter.json")))
    assert len(cases) >= 9
    for c in cases:
        queries, targets = _prefilter_case(c)
        assert sum(len(t) for t in targets) == c["cells"]
        for chunks in (1, 3):                                  # the same lists whether the database arrives whole or in chunks
            pf = oracle.Prefilter(queries, c["k"], c["max_candidates"])
            step = (len(targets) + chunks - 1) // chunks
            for o in range(0, len(targets), step):
                pf.add(targets[o:o + step])
            ids, counts, scores, cut = pf.finish()
            if c.get("digest"):
                import make_prefilter_golden as g
                got = [[int(x) for x in ids[q, :counts[q]]] for q in range(c["nq"])]
                assert [len(x) for x in got] == c["list_sizes"] and g.lists_digest(got) == c["lists_sha256"], c["name"]
                continue
            for q, want in enumerate(c["lists"]):
                got = [int(x) for x in ids[q, :counts[q]]]
                assert got == sorted(got)
                if "exact" in want:
                    assert got == want["exact"], (c["name"], q)
                else:
                    assert len(got) == want["size"], (c["name"], q)
                    assert set(want["must"]) <= set(got), (c["name"], q)
                    assert (scores[q, :counts[q]] >= cut[q]).all()


def test_prefilter_hand_cases():
    """Small cases worked by hand from the reference's text."""
    enc = oracle.encode
    # one shared 5-mer -> LIS 1; score 1 / len(target)
    pf = oracle.Prefilter([enc("AAAAACDEFGAAAAA")], 5, 10)
    L = pf.lis([enc("WWWCDEFGWWW"), enc("CDEF"), enc("WWWWWWWW")])
    assert L.tolist() == [[1, 0, 0]]
    # consecutive equal k-mers of the TARGET are looked up once (database_search.cpp:212): poly-A target of 30 has 26 k-mers,
    # all equal -> one lookup, which hits every poly-A position of the query: positions 0..5 ascending -> LIS 6
    pf = oracle.Prefilter([enc("AAAAAAAAAA")], 5, 10)              # 6 k-mers, all AAAAA, positions 0..5
    assert pf.lis([enc("A" * 30)]).tolist() == [[6]]
    # ... but an alternating repeat is not skipped: target k-mers QAQAQ, AQAQA, QAQAQ, ... each hit all matching positions
    pf = oracle.Prefilter([enc("QAQAQAQA")], 5, 10)                # k-mers: QAQAQ@0, AQAQA@1, QAQAQ@2, AQAQA@3
    # target QAQAQAQ: k-mers QAQAQ, AQAQA, QAQAQ -> hit list [0,2, 1,3, 0,2] -> strict LIS = 0,1,2 or 0,2,3 -> 3
    assert pf.lis([enc("QAQAQAQ")]).tolist() == [[3]]
    # the LIS is strict and taken over query positions in target order: a reversed arrangement scores 1 per block
    q = "ACDEFGHIKLMNPQRSTVWY"
    pf = oracle.Prefilter([enc(q)], 5, 10)
    assert pf.lis([enc(q)]).tolist() == [[16]]                     # 16 k-mers in order
    assert pf.lis([enc(q[10:] + q[:10])]).tolist() == [[6]]        # two blocks of 6 k-mers each, second block lower -> 6
    # ties at the cut go to the lower database index; the score is a float division
    pf = oracle.Prefilter([enc("CDEFGHIK")], 5, 2)
    pf.add([enc("WWCDEFGWW"), enc("WWDEFGHWW"), enc("WWEFGHIWW"), enc("CDEFGHIK")])
    ids, counts, scores, cut = pf.finish()
    assert counts.tolist() == [2] and ids[0, :2].tolist() == [0, 3]
    assert scores[0, 0] == np.float32(1) / np.float32(9) and scores[0, 1] == np.float32(4) / np.float32(8)
    assert cut[0] == np.float32(1) / np.float32(9)
    # k outside 3..5 is refused (main.cpp:169)
    with pytest.raises(ValueError):
        oracle.Prefilter([enc("ACDEFG")], 6, 10)
