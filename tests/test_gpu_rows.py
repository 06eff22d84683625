"""One small test per row of SURVEY.md 8 (the hot-path scope table) and per BASELINE.json config, collected FIRST (tests/conftest.py).

The driver runs `pytest -m gpu -x`: one failure anywhere hides every test behind it (GPUTEST_r05 stopped at its fourth test and left
fifteen rows without evidence).  These tests put the row evidence on the first screen of the log -- the test id names the row -- each one
a HIP-path-vs-oracle (or vs tests/golden) EQUALITY through the C ABI at a size that takes seconds; the files behind this one go deeper
into every row (routes, tiers, edge cases, full sizes).  No assertion here reads a clock, a rate or a ratio.

  a-R  the recurrence            a1  sift4g's wrapper loop      a2  swsharp alignDatabase steps 1-4    a3  candidate lists
  a4   valueFunction -> eValues  a5  createEValueParams         a6  Scorer (built-in tables)           a7  Chain codes 0..25
  a8   ChainDatabase             a9  dbAlignmentsMerge          a10 DbAlignment accessors              b   the C ABI on the GPU box
  c    golden vectors on the HIP path   d  what the bench counts      e  shard + gather + merge (one process)
  f1   device top-k / e-value cut       f2 traceback                  f3 k-mer prefilter                f4 packed database file
  dagger  the reference CLI's .SIFTprediction bytes       cfg0..cfg4  BASELINE.json configs (cfg1..cfg4 at reduced, routed-alike sizes)
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, rand_seqs

pytestmark = pytest.mark.gpu

MOVE = {0: 1, 1: 2, 2: 3}        # oracle move codes -> swsharp's MOVE_DIAG / MOVE_LEFT / MOVE_UP


@pytest.fixture(scope="module")
def s4g():
    import sift4g_amd
    assert sift4g_amd.device_count() >= 1
    return sift4g_amd


@pytest.fixture(scope="module")
def scorer(s4g):
    return s4g.Scorer("BLOSUM_62", 10, 1)


@pytest.fixture(scope="module")
def sw():
    import swsharp_ctypes
    return swsharp_ctypes


def _homolog_set(seed, nq, nt, qlen=(60, 220), tlen=(30, 260), every=7, alphabet=20):
    """random queries and targets, a mutated copy (substitutions + one indel) of a query at every `every`-th target"""
    rng = np.random.default_rng(seed)
    qs = rand_seqs(rng, nq, *qlen, alphabet)
    ts = rand_seqs(rng, nt, *tlen, alphabet)
    for k in range(0, nt, every):
        m = qs[(k // every) % nq].copy()
        idx = rng.choice(len(m), size=len(m) // 4, replace=False)
        m[idx] = rand_seqs(rng, 1, len(idx), len(idx), alphabet)[0]
        cut = int(rng.integers(10, len(m) - 10))
        ts[k] = np.concatenate([m[:cut], rand_seqs(rng, 1, 3, 3, alphabet)[0], m[cut + 2:]])
    return rng, qs, ts


def _hits(L, sw, out, n):
    return [(L.dbAlignmentGetTargetIdx(out[k]), L.dbAlignmentGetScore(out[k]), L.dbAlignmentGetValue(out[k]),
             L.dbAlignmentGetQueryStart(out[k]), L.dbAlignmentGetQueryEnd(out[k]), L.dbAlignmentGetTargetStart(out[k]), L.dbAlignmentGetTargetEnd(out[k]),
             tuple(sw.path_of(L.dbAlignmentGetPathLen, L.dbAlignmentGetMove, out[k]))) for k in range(n)]


def _expected_hits(oracle_mod, M, er, q, ts, row, cells, P, max_al, thr, idx_of=lambda t: t):
    """steps 2-4 on ORACLE results only (no library code): e-value of every score, threshold, order (value asc, score desc, index asc),
    best max_al, traceback by the oracle."""
    vals = [er.evalue(int(row[t]), len(q), len(ts[t]), cells, P) for t in range(len(ts))]
    order = sorted((t for t in range(len(ts)) if vals[t] <= thr), key=lambda t: (vals[t], -int(row[t]), t))[:max_al]
    out = []
    for t in order:
        a = oracle_mod.sw_align(q, ts[t], M, 10, 1)
        assert a["score"] == int(row[t])
        out.append((idx_of(t), int(row[t]), vals[t], a["q_start"], a["q_end"], a["t_start"], a["t_end"], tuple(MOVE[m] for m in a["path"])))
    return out


def _same_hits(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[:2] == w[:2] and g[3:] == w[3:], (g[:7], w[:7])
        assert g[2] == pytest.approx(w[2], rel=1e-12, abs=0.0)


# ---------------------------------------------------------------------------------------------------------------- a-R

def test_aR_recurrence_int32_scores_equal_the_oracle(s4g, oracle_mod, blosum62, micro, scorer):
    """SURVEY 8a a-R: affine-gap local scores, int32 out -- the hand-checked micro cases (gap of length k costs open + (k-1) ext,
    all-negative -> 0), then every letter code on a ragged set, on the latency route, the packed scan tiers and the int32 kernel."""
    qs = [oracle_mod.encode(c["q"]) for c in micro["cases"]]
    ts = [oracle_mod.encode(c["t"]) for c in micro["cases"]]
    for mode in (dict(), dict(no_pairs=True), dict(force_i32=True, no_pairs=True)):
        got = s4g.scan_host(s4g.Database(ts), s4g.QuerySet(qs), scorer, **mode)
        for i, c in enumerate(micro["cases"]):
            assert got[i, i] == c["score"], (mode, c["name"])
    rng = np.random.default_rng(601)
    qs, ts = rand_seqs(rng, 7, 1, 260, 26), rand_seqs(rng, 333, 1, 420, 26)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    for mode in (dict(), dict(no_pairs=True), dict(no_pairs=True, no_drift=True), dict(no_pairs=True, no_f16=True), dict(no_pairs=True, force_i32=True),
                 dict(no_pairs=True, force_i32=True, no_drift=True), dict(block_pipe=True)):
        got = s4g.scan_host(db, q, scorer, **mode)
        assert got.dtype == np.int32 and np.array_equal(got, exp), mode
    exp = oracle_mod.scan(qs, ts, blosum62, 5, 5, threads=8)                # open == extend (main.cpp:297-303: extend <= open)
    assert np.array_equal(s4g.scan_host(db, q, s4g.Scorer("BLOSUM_62", 5, 5), no_pairs=True), exp)


# ---------------------------------------------------------------------------------------------------------------- a1

def test_a1_wrapper_loop_over_chunks_equals_one_pass_on_oracle_results(s4g, sw, oracle_mod, blosum62):
    """database_alignment.cpp:21-127 replayed through the C API: the database in three chunks, per (query, chunk) a candidate database
    created, one 15-argument alignDatabase call, target indices read (:89), the database deleted (:92), dbAlignmentsMerge per chunk
    (:97-104).  Result == steps 2-4 computed from ORACLE scores, e-values (oracle/evalue_ref.py) and tracebacks over the whole set."""
    from oracle import evalue_ref as er
    L, ci, pvp = sw.L, sw.ci, sw.pvp
    rng, qs, ts = _homolog_set(611, 4, 360)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    cells = int(sum(len(t) for t in ts))
    s = sw.scorer(10, 1)
    tch = [sw.chain(f"t{i}", t) for i, t in enumerate(ts)]
    qch = [sw.chain(f"q{i}", q) for i, q in enumerate(qs)]
    ev, cb, cards, K, thr = L.createEValueParams(cells, s), sw.evalue_callback(), (ci * 1)(0), 15, 1e-3
    cand = [sorted(rng.choice(len(ts), size=200, replace=False).tolist() + list(range(0, len(ts), 7))[:40]) for _ in qs]
    cand = [sorted(set(c)) for c in cand]
    total, lens, keep = None, None, []
    for lo, hi in ((0, 120), (120, 240), (240, 360)):
        part, plens = (pvp * len(qch))(), (ci * len(qch))()
        for qi, q in enumerate(qch):
            used = [t for t in cand[qi] if lo <= t < hi]                     # createFilteredDatabase (:136-162)
            arr = sw.chain_array([tch[t] for t in used])
            keep.append(arr)
            cdb = L.chainDatabaseCreate(arr, 0, len(used), cards, 1)
            out, n = pvp(), ci()
            L.alignDatabase(ctypes.byref(out), ctypes.byref(n), 0, q, cdb, s, K, cb, ev, thr, None, 0, cards, 1, None)
            assert all(0 <= L.dbAlignmentGetTargetIdx(out[k]) < len(used) for k in range(n.value))
            L.chainDatabaseDelete(cdb)
            part[qi], plens[qi] = out, n.value
        if total is None:
            total, lens = part, plens
        else:
            L.dbAlignmentsMerge(total, lens, part, plens, len(qch), K)
    P = er.gumbel_parameters("BLOSUM_62", 10, 1, "swsharp")
    n_hits = 0
    for qi in range(len(qs)):
        sub = [ts[t] for t in cand[qi]]
        want = _expected_hits(oracle_mod, blosum62, er, qs[qi], sub, exp[qi, cand[qi]], cells, P, K, thr)
        got = [(L.chainGetName(L.dbAlignmentGetTarget(total[qi][k])).decode(),) + h[1:] for k, h in enumerate(_hits(L, sw, total[qi], lens[qi]))]
        # (names instead of indices: a chunk's indices are relative to its own candidate database; compared as sets ordered by score and
        # name, since equal (value, score) keys coming from different chunks may sit either way round after the merges)
        _same_keys(got, [(f"t{cand[qi][w[0]]}",) + w[1:] for w in want])
        n_hits += lens[qi]
        for k in range(lens[qi]):
            L.dbAlignmentDelete(total[qi][k])
    assert n_hits >= 20
    L.deleteEValueParams(ev); L.scorerDelete(s)


def _same_keys(got, want):
    """equal up to float rounding of the e-value (1e-12 relative) -- names, scores, coordinates and paths exactly"""
    g = sorted(got, key=lambda h: (-h[1], h[0])); w = sorted(want, key=lambda h: (-h[1], h[0]))
    assert len(g) == len(w)
    for a, b in zip(g, w):
        assert a[:2] == b[:2] and a[3:] == b[3:], (a[:7], b[:7])
        assert a[2] == pytest.approx(b[2], rel=1e-12, abs=0.0)
    return True


# ---------------------------------------------------------------------------------------------------------------- a2 + cfg0

def test_a2_cfg0_align_database_steps_1_to_4_on_the_reference_fixture(s4g, sw, oracle_mod, blosum62, cfg1):
    """BASELINE configs[0] (test_files/query.fasta x sample_protein_database.fa) through the 15-argument call of
    database_alignment.cpp:83-86: step 1 == tests/golden/cfg1_scores.json, steps 2-4 == e-values, cut, order and tracebacks computed
    from the oracle alone; and the `indexes` subset form."""
    from oracle import evalue_ref as er
    L, ci, pvp = sw.L, sw.ci, sw.pvp
    s = sw.scorer(10, 1)
    ts = [c for _, c in cfg1["targets"]]
    targets = [sw.chain(n, c) for n, c in cfg1["targets"]]
    arr = sw.chain_array(targets)
    cards = (ci * 1)(0)
    cdb = L.chainDatabaseCreate(arr, 0, len(targets), cards, 1)
    cells = int(sum(len(t) for t in ts))
    assert cells == 3091
    ev, cb = L.createEValueParams(cells, s), sw.evalue_callback()
    P = er.gumbel_parameters("BLOSUM_62", 10, 1, "swsharp")
    for qi, (qn, qc) in enumerate(cfg1["queries"]):
        q = sw.chain(qn, qc)
        out, n = pvp(), ci()
        L.alignDatabase(ctypes.byref(out), ctypes.byref(n), 0, q, cdb, s, 400, cb, ev, 1e-4, None, 0, cards, 1, None)
        want = _expected_hits(oracle_mod, blosum62, er, qc, ts, cfg1["scores"][qi], cells, P, 400, 1e-4)
        assert n.value >= 5
        _same_hits(_hits(L, sw, out, n.value), want)
        idx = (ci * 3)(4, 1, 7)
        sub, ns = pvp(), ci()
        L.alignDatabase(ctypes.byref(sub), ctypes.byref(ns), 0, q, cdb, s, 400, cb, ev, 1e9, idx, 3, cards, 1, None)
        got = _hits(L, sw, sub, ns.value)
        assert sorted(h[0] for h in got) == [1, 4, 7] and all(cfg1["scores"][qi][h[0]] == h[1] for h in got)
        L.chainDelete(q)
    L.deleteEValueParams(ev); L.chainDatabaseDelete(cdb); L.scorerDelete(s)


def test_cfg0_fixture_scores_on_every_route(s4g, cfg1, scorer):
    """BASELINE configs[0] on the HIP path: the 2 x 9 golden matrix on the latency route, the scan kernels and the int32 kernel."""
    db, q = s4g.Database([c for _, c in cfg1["targets"]]), s4g.QuerySet([c for _, c in cfg1["queries"]])
    for mode in (dict(), dict(no_pairs=True), dict(no_pairs=True, force_i32=True)):
        assert np.array_equal(s4g.scan_host(db, q, scorer, **mode), cfg1["scores"]), mode


# ---------------------------------------------------------------------------------------------------------------- a3

def test_a3_candidate_lists_equal_per_query_databases(s4g, sw, oracle_mod, blosum62):
    """createFilteredDatabase (database_alignment.cpp:136-162) as a pair list: shotgunDatabaseCandidates with sorted per-query id lists
    (unequal sizes, an empty one, a NULL one, the whole database) == steps 2-4 from the oracle on just those chains, target indices
    into the big database."""
    from oracle import evalue_ref as er
    L, ci, pvp = sw.L, sw.ci, sw.pvp
    rng, qs, ts = _homolog_set(621, 6, 300, every=6)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    cells = int(sum(len(t) for t in ts))
    s = sw.scorer(10, 1)
    tch, qch = [sw.chain(f"t{i}", t) for i, t in enumerate(ts)], [sw.chain(f"q{i}", q) for i, q in enumerate(qs)]
    tarr, qarr = sw.chain_array(tch), sw.chain_array(qch)
    ev, cb = L.createEValueParams(cells, s), sw.evalue_callback()
    lists = [sorted(rng.choice(len(ts), size=int(n), replace=False).tolist()) for n in (120, 7, 0, 300, 65, 1)]
    arrs = [(ci * max(len(l), 1))(*l) for l in lists]
    ptrs = (ctypes.POINTER(ci) * len(lists))(*[ctypes.cast(a, ctypes.POINTER(ci)) for a in arrs])
    ptrs[2] = ctypes.POINTER(ci)()
    lens_in = (ci * len(lists))(*[len(l) for l in lists])
    cdb = L.chainDatabaseCreate(tarr, 0, len(tch), None, 0)
    outs, lens = ctypes.POINTER(pvp)(), ctypes.POINTER(ci)()
    L.shotgunDatabaseCandidates(ctypes.byref(outs), ctypes.byref(lens), 0, qarr, len(qch), cdb, s, 10, cb, ev, 1e-3, ptrs, lens_in, None, 0, None)
    P = er.gumbel_parameters("BLOSUM_62", 10, 1, "swsharp")
    total = 0
    for i, l in enumerate(lists):
        if not l:
            assert lens[i] == 0
            continue
        want = _expected_hits(oracle_mod, blosum62, er, qs[i], [ts[t] for t in l], exp[i, l], cells, P, 10, 1e-3, idx_of=lambda k, l=l: l[k])
        _same_hits(_hits(L, sw, outs[i], lens[i]), want)
        total += lens[i]
    assert total >= 10
    L.deleteShotgunDatabase(outs, lens, len(qs)); L.chainDatabaseDelete(cdb); L.deleteEValueParams(ev); L.scorerDelete(s)


# ---------------------------------------------------------------------------------------------------------------- a4, a5

def test_a4_value_function_evalues_of_device_scores(s4g, sw, oracle_mod, blosum62):
    """valueFunction -> eValues (database_alignment.cpp:129-134): the callback the library invokes sees the DEVICE's scores (== oracle's)
    for exactly the targets of the call, and what it returns is what the alignments carry; the values == oracle/evalue_ref.py (1e-12)."""
    from oracle import evalue_ref as er
    L, ci, cd, pvp = sw.L, sw.ci, sw.cd, sw.pvp
    rng, qs, ts = _homolog_set(631, 2, 150)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    cells = int(sum(len(t) for t in ts))
    s = sw.scorer(10, 1)
    tch = [sw.chain(f"t{i}", t) for i, t in enumerate(ts)]
    arr = sw.chain_array(tch)
    cdb = L.chainDatabaseCreate(arr, 0, len(tch), None, 0)
    ev = L.createEValueParams(cells, s)
    P = er.gumbel_parameters("BLOSUM_62", 10, 1, "swsharp")
    seen = []

    def fn(values, scores, query, database, n, cards, cards_len, param):
        seen.append([scores[i] for i in range(n)])
        L.eValues(values, scores, query, database, n, cards, cards_len, param)
    cb = sw.ValueFunction(fn)
    for qi, qc in enumerate(qs):
        seen.clear()
        q = sw.chain(f"q{qi}", qc)
        out, n = pvp(), ci()
        L.alignDatabase(ctypes.byref(out), ctypes.byref(n), 0, q, cdb, s, 400, cb, ev, 1e9, None, 0, None, 0, None)
        assert sorted(sum(seen, [])) == sorted(int(x) for x in exp[qi])          # every target's score passed through the callback once
        assert n.value == min(400, len(ts))
        for h in _hits(L, sw, out, n.value):
            assert h[1] == int(exp[qi, h[0]])
            assert h[2] == pytest.approx(er.evalue(h[1], len(qc), len(ts[h[0]]), cells, P), rel=1e-12, abs=0.0)
        L.chainDelete(q)
    L.deleteEValueParams(ev); L.chainDatabaseDelete(cdb); L.scorerDelete(s)


def test_a5_evalue_params_known_answers(sw):
    """createEValueParams(cells, scorer) (main.cpp:210) + eValues against tests/golden/evalue_kat.json (oracle/evalue_ref.py's vectors:
    eight matrices, scores 1..12 000, database sizes from the fixture's 3 091 residues to configs[3]'s 5.1e10), on the GPU box."""
    L, ci, cd = sw.L, sw.ci, sw.cd
    kat = json.load(open(os.path.join(GOLDEN, "evalue_kat.json")))
    groups = {}
    for c in kat["cases"]:
        if c[3] == "swsharp":                       # the default convention (the other one needs an environment switch: tests/test_swsharp_host.py)
            groups.setdefault(tuple(c[:7]), []).append((c[7], c[8]))
    assert len(groups) >= 20
    for (matrix, go, ge, conv, m, n, db), rows in groups.items():
        s = sw.scorer(go, ge, matrix)
        ev = L.createEValueParams(db, s)
        q, t = sw.chain("q", np.zeros(m, dtype=np.uint8)), sw.chain("t", np.zeros(n, dtype=np.uint8))
        vals = (cd * len(rows))()
        L.eValues(vals, (ci * len(rows))(*[r[0] for r in rows]), q, sw.chain_array([t] * len(rows)), len(rows), None, 0, ev)
        for (score, want), got in zip(rows, vals):
            assert got == pytest.approx(want, rel=1e-12, abs=0.0), (matrix, go, ge, m, n, db, score)
        L.deleteEValueParams(ev); L.chainDelete(q); L.chainDelete(t); L.scorerDelete(s)


# ---------------------------------------------------------------------------------------------------------------- a6, a7, a8

# (six of the eight tables here, so that the rows and configs of this file are the first thirty tests of the suite; all eight -- and the
# CLI's alias names -- in tests/test_gpu_parity.py::test_every_builtin_matrix and tests/test_oracle.py)
@pytest.mark.parametrize("name,go,ge", [("BLOSUM_62", 10, 1), ("BLOSUM_45", 14, 2), ("BLOSUM_80", 10, 1), ("PAM_30", 9, 1), ("PAM_70", 10, 1), ("PAM_250", 14, 2)])
def test_a6_scorer_every_builtin_table(s4g, sw, oracle_mod, name, go, ge):
    """scorerCreateMatrix(&s, name, open, extend) (main.cpp:208, names :304-314): the table the host API answers with == the table the
    device scores with == tests/golden/matrix_sha256.json, and a HIP scan under it == the oracle under the same table."""
    from sift4g_amd import device as s4gd
    L = sw.L
    M = np.ascontiguousarray(s4gd.matrix_by_name(name), dtype=np.int32)
    s = sw.scorer(go, ge, name)
    assert (L.scorerGetGapOpen(s), L.scorerGetGapExtend(s)) == (go, ge)
    host = np.array([[L.scorerScore(s, bytes([a]), bytes([b])) for b in range(26)] for a in range(26)], dtype=np.int32)
    assert np.array_equal(host, M) and L.scorerGetMaxScore(s) == int(M.max())
    L.scorerDelete(s)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import matrix_digest as md
    assert md.digest(md.canonical_from_table26(M)) == json.load(open(os.path.join(GOLDEN, "matrix_sha256.json")))[name]
    rng = np.random.default_rng(641)
    qs, ts = rand_seqs(rng, 4, 20, 200, 26), rand_seqs(rng, 150, 1, 300, 26)
    ts[3] = qs[1].copy(); ts[77] = qs[2][5:150].copy()
    exp = oracle_mod.scan(qs, ts, M, go, ge, threads=8)
    db, q, sc = s4g.Database(ts), s4g.QuerySet(qs), s4g.Scorer(name, go, ge)
    for mode in (dict(), dict(no_pairs=True), dict(no_pairs=True, force_i32=True)):
        assert np.array_equal(s4g.scan_host(db, q, sc, **mode), exp), (name, mode)


def test_a7_chain_codes_are_letter_minus_A(s4g, sw, oracle_mod, blosum62, scorer):
    """Chain (hash.cpp:16-19,25-36: 5-bit codes 0..25): chainCreate keeps letters (upper-cased), chainGetCodes = letter - 'A',
    chainGetChar / chainGetLength / chainGetName / views agree; the 26 codes scored on the device == the oracle."""
    L = sw.L
    text = b"ACDEFGHIKLMNPQRSTVWYBJOUXZacdxyz"
    c = L.chainCreate(b"every letter", 12, text, len(text))
    n = L.chainGetLength(c)
    assert n == len(text) and L.chainGetName(c) == b"every letter"
    codes = sw.chain_codes(c)
    assert codes == [ch - 65 for ch in text.upper()] and max(codes) == 25 and min(codes) == 0
    assert bytes(ord(L.chainGetChar(c, i)) for i in range(n)) == text.upper()
    v = L.chainCreateView(c, 3, 10, 0)                                          # inclusive bounds (sift_prediction.cpp:197)
    assert L.chainGetLength(v) == 8 and sw.chain_codes(v) == codes[3:11]
    r = L.chainCreateView(c, 3, 10, 1)
    assert sw.chain_codes(r) == codes[3:11][::-1]
    qs = [np.array(codes, dtype=np.uint8), np.arange(26, dtype=np.uint8)]
    ts = [np.array(codes[::-1], dtype=np.uint8), np.tile(np.arange(26, dtype=np.uint8), 3), np.array(codes, dtype=np.uint8)]
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1)
    for mode in (dict(), dict(no_pairs=True)):
        assert np.array_equal(s4g.scan_host(s4g.Database(ts), s4g.QuerySet(qs), scorer, **mode), exp)
    for x in (r, v, c):
        L.chainDelete(x)


def test_a8_chain_database_windows_and_reuse(s4g, sw, oracle_mod, blosum62):
    """chainDatabaseCreate(chains, start, len, cards, n) / chainDatabaseDelete (database_alignment.cpp:80-81,92): a window [start,
    start + len) of the array -- target indices are relative to the window -- created and deleted forty times over (the per-device chain
    arena is recycled) with scores == the oracle every time."""
    L, ci, pvp = sw.L, sw.ci, sw.pvp
    rng, qs, ts = _homolog_set(651, 2, 200)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    s = sw.scorer(10, 1)
    tch = [sw.chain(f"t{i}", t) for i, t in enumerate(ts)]
    arr = sw.chain_array(tch)
    ev, cb = L.createEValueParams(int(sum(len(t) for t in ts)), s), sw.evalue_callback()
    qch = [sw.chain(f"q{i}", q) for i, q in enumerate(qs)]
    for it in range(40):
        start = int(rng.integers(0, 150)); n = int(rng.integers(1, 200 - start + 1))
        cdb = L.chainDatabaseCreate(arr, start, n, None, 0)
        qi = it % 2
        out, cnt = pvp(), ci()
        L.alignDatabase(ctypes.byref(out), ctypes.byref(cnt), 0, qch[qi], cdb, s, 1000, cb, ev, 1e9, None, 0, None, 0, None)
        assert cnt.value == n
        got = {L.dbAlignmentGetTargetIdx(out[k]): L.dbAlignmentGetScore(out[k]) for k in range(cnt.value)}
        assert got == {k: int(exp[qi, start + k]) for k in range(n)}
        assert all(L.chainGetName(L.dbAlignmentGetTarget(out[k])) == f"t{start + L.dbAlignmentGetTargetIdx(out[k])}".encode() for k in range(cnt.value))
        L.chainDatabaseDelete(cdb)
        for k in range(cnt.value):
            L.dbAlignmentDelete(out[k])
        sw.libc.free(out)
    L.deleteEValueParams(ev); L.scorerDelete(s)


# ---------------------------------------------------------------------------------------------------------------- a9, a10

def test_a9_merge_of_device_lists_equals_the_whole_database_call(s4g, sw, oracle_mod, blosum62):
    """dbAlignmentsMerge (database_alignment.cpp:101-102): per-query lists from two halves of the database (device-scored, target
    names as identity) merged with a cap -> the list one call over the whole database gives; the source lengths are zeroed."""
    L, ci, pvp = sw.L, sw.ci, sw.pvp
    rng, qs, ts = _homolog_set(661, 3, 240, every=3)
    s = sw.scorer(10, 1)
    tch, qch = [sw.chain(f"t{i:03d}", t) for i, t in enumerate(ts)], [sw.chain(f"q{i}", q) for i, q in enumerate(qs)]
    ev, cb = L.createEValueParams(int(sum(len(t) for t in ts)), s), sw.evalue_callback()
    K = 25

    def run(lo, hi):
        arr = sw.chain_array(tch[lo:hi])
        cdb = L.chainDatabaseCreate(arr, 0, hi - lo, None, 0)
        lists, lens = (pvp * len(qch))(), (ci * len(qch))()
        for qi, q in enumerate(qch):
            out, n = pvp(), ci()
            L.alignDatabase(ctypes.byref(out), ctypes.byref(n), 0, q, cdb, s, K, cb, ev, 10.0, None, 0, None, 0, None)
            lists[qi], lens[qi] = out, n.value
        L.chainDatabaseDelete(cdb)
        return lists, lens, arr
    key = lambda a: (L.dbAlignmentGetValue(a), -L.dbAlignmentGetScore(a), L.chainGetName(L.dbAlignmentGetTarget(a)))
    whole, wl, _k0 = run(0, 240)
    a, al, _k1 = run(0, 130)
    b, bl, _k2 = run(130, 240)
    L.dbAlignmentsMerge(a, al, b, bl, len(qch), K)
    for qi in range(len(qch)):
        assert al[qi] == wl[qi] == K and bl[qi] == 0
        got, want = [key(a[qi][k]) for k in range(K)], [key(whole[qi][k]) for k in range(K)]
        assert [g[:2] for g in got] == [w[:2] for w in want] and sorted(got) == sorted(want)
        assert all(got[k][:2] <= got[k + 1][:2] for k in range(K - 1))
    L.deleteEValueParams(ev); L.scorerDelete(s)


def test_a10_db_alignment_accessors_equal_the_oracle_traceback(s4g, sw, oracle_mod, blosum62):
    """dbAlignmentGetTargetIdx / GetTarget / GetQueryStart / GetPathLen / ToAlignment (database_alignment.cpp:89;
    select_alignments.cpp:145-152,244-300): coordinates and the move string of every returned alignment == the oracle's traceback
    (MOVE_LEFT = gap in the query, MOVE_UP = gap in the target), and the Alignment view repeats them."""
    L, ci, pvp = sw.L, sw.ci, sw.pvp
    rng, qs, ts = _homolog_set(671, 3, 180, every=4)
    s = sw.scorer(10, 1)
    tch = [sw.chain(f"t{i}", t) for i, t in enumerate(ts)]
    arr = sw.chain_array(tch)
    cdb = L.chainDatabaseCreate(arr, 0, len(tch), None, 0)
    ev, cb = L.createEValueParams(int(sum(len(t) for t in ts)), s), sw.evalue_callback()
    moves = set()
    for qi, qc in enumerate(qs):
        q = sw.chain(f"q{qi}", qc)
        out, n = pvp(), ci()
        L.alignDatabase(ctypes.byref(out), ctypes.byref(n), 0, q, cdb, s, 30, cb, ev, 1.0, None, 0, None, 0, None)
        assert n.value >= 10
        for k in range(n.value):
            a = out[k]
            t = L.dbAlignmentGetTargetIdx(a)
            o = oracle_mod.sw_align(qc, ts[t], blosum62, 10, 1)
            assert L.chainGetName(L.dbAlignmentGetTarget(a)) == f"t{t}".encode() and L.dbAlignmentGetScore(a) == o["score"]
            assert (L.dbAlignmentGetQueryStart(a), L.dbAlignmentGetQueryEnd(a), L.dbAlignmentGetTargetStart(a), L.dbAlignmentGetTargetEnd(a)) == \
                   (o["q_start"], o["q_end"], o["t_start"], o["t_end"])
            path = sw.path_of(L.dbAlignmentGetPathLen, L.dbAlignmentGetMove, a)
            assert path == [MOVE[m] for m in o["path"]]
            moves |= set(path)
            al = L.dbAlignmentToAlignment(a)
            assert (L.alignmentGetQueryStart(al), L.alignmentGetTargetStart(al), L.alignmentGetPathLen(al), L.alignmentGetScore(al)) == \
                   (o["q_start"], o["t_start"], len(path), o["score"])
            assert sw.path_of(L.alignmentGetPathLen, L.alignmentGetMove, al) == path
            L.alignmentDelete(al)
        L.chainDelete(q)
    assert moves == {1, 2, 3}
    L.deleteEValueParams(ev); L.chainDatabaseDelete(cdb); L.scorerDelete(s)


# ---------------------------------------------------------------------------------------------------------------- b, c, d

def test_b_c_abi_on_the_gpu_box_and_loud_failure_without_it(s4g):
    """the product is the C-ABI library -- every symbol of include/s4g_hip.h resolves in the loaded libs4g_hip.so, it sees the
    GPU, and the Python mirror refuses to work without it (no CPU fallback: a child process pointed at a missing library must raise)."""
    import re
    from sift4g_amd._lib import lib
    hdr = open(os.path.join(ROOT, "include", "s4g_hip.h")).read()
    names = sorted(set(re.findall(r"\b(s4g_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), n
    assert lib.s4g_abi_version() >= 6 and s4g.device_count() >= 1
    code = "import os, sys; sys.path.insert(0, %r); os.environ['S4G_LIB'] = '/nonexistent/libs4g_hip.so'\ntry:\n    import sift4g_amd\nexcept Exception as e:\n    print('REFUSED', type(e).__name__)\n" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFUSED" in r.stdout, (r.stdout, r.stderr[-500:])


def test_c_oracle_golden_vectors_on_the_hip_path(s4g, oracle_mod, micro, cfg1, scorer):
    """the committed vectors (tests/golden/micro.json, cfg1_scores.json: self-score identities of the fixture) hold on the HIP
    path exactly as tests/test_oracle.py pins them on the oracle."""
    qs = [oracle_mod.encode(c["q"]) for c in micro["cases"]]
    ts = [oracle_mod.encode(c["t"]) for c in micro["cases"]]
    got = s4g.scan_host(s4g.Database(ts), s4g.QuerySet(qs), scorer, no_pairs=True)
    assert [int(got[i, i]) for i in range(len(qs))] == [c["score"] for c in micro["cases"]]
    from sift4g_amd import device as s4gd
    M = s4gd.matrix_by_name("BLOSUM_62")
    tq = [c for _, c in cfg1["queries"]]
    self_scores = s4g.scan_host(s4g.Database(tq), s4g.QuerySet(tq), scorer, no_pairs=True)
    for i, c in enumerate(tq):
        assert self_scores[i, i] == int(M[c, c].sum())                    # a sequence against itself: the sum of its diagonal entries


def test_d_what_the_bench_counts(s4g, scorer):
    """GCUPS = sum of Lq x Lt over the scored pairs -- s4g_scan_cells is that sum, s4g_scan_plan names the kernel a call of
    the bench's shape runs, s4g_scan_stats counts every query in exactly one tier.  (Presence and arithmetic only: no rate is asserted.)"""
    import torch
    from sift4g_amd import device as s4gd, synth
    cfg = synth.make_config(2, n_queries=20, n_targets=3000, homologs_per_query=4)
    db, q = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"]), s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    ql, tl = np.diff(cfg["q_off"].astype(np.int64)), np.diff(cfg["t_off"].astype(np.int64))
    assert s4gd.scan_cells(db, q) == int(ql.sum()) * int(tl.sum())
    assert s4gd.scan_cells(db, q, 3, 5) == int(ql[3:8].sum()) * int(tl.sum())
    plan = s4gd.scan_plan(db, q, scorer)
    assert plan["kernel"].startswith("sw_scan_") and plan["grid_blocks"] >= 1
    out = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, out)
    torch.cuda.synchronize()
    st = s4gd.scan_stats(0)
    assert st["tier_drift"] + st["tier_m3"] + st["tier_i16"] + st["tier_i32"] == q.n
    assert s4gd.scan_clock(0)["launches"] >= 1


# ---------------------------------------------------------------------------------------------------------------- e

def test_e_shards_gather_and_merge_equal_one_scan(s4g, oracle_mod, blosum62, scorer):
    """SURVEY 8e in one process on one GPU: (query-shard) the lists of two query ranges concatenated, (DB-shard) the lists of two
    length-balanced database slices merged by the device kernel behind s4g_gather_topk and by its host twin -- both == the top-k of one
    scan, which == the sorted oracle scores; then the one-rank RCCL communicator hands back what went in."""
    import torch
    from sift4g_amd import shard
    from sift4g_amd._lib import check, lib
    rng, qs, ts = _homolog_set(681, 6, 700, every=5)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    K = 30
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    ids, sc, cnt = [x.cpu() for x in s4g.scan_topk(db, q, scorer, K)]
    for i in range(len(qs)):
        order = sorted((j for j in range(len(ts)) if exp[i, j] >= 1), key=lambda j: (-exp[i, j], j))[:K]
        assert ids[i].tolist() == order and sc[i].tolist() == [int(exp[i, j]) for j in order] and int(cnt[i]) == K
    # query-shard: ranges [0, 3) and [3, 6)
    out = torch.empty((3, db.n), dtype=torch.int32, device="cuda:0")
    parts = []
    for b in (0, 3):
        s4g.scan(db, q, scorer, out, q_begin=b, q_count=3)
        parts.append([x.cpu() for x in s4g.topk_device(db, out, K, 1)])
    assert torch.equal(torch.cat([p[0] for p in parts]), ids) and torch.equal(torch.cat([p[1] for p in parts]), sc)
    # DB-shard: two length-balanced slices, local ids -> global ids, K-way merge on the device and on the host
    lengths = np.array([len(t) for t in ts])
    l_ids, l_sc = [], []
    for mine in shard.balanced_db_shards(lengths, 2):
        mine = mine.numpy()
        part = s4g.Database([ts[i] for i in mine], tile_seqs=256)
        pi, ps, pc = [x.cpu() for x in s4g.scan_topk(part, q, scorer, K)]
        l_ids.append(torch.where(pi >= 0, torch.from_numpy(mine.astype(np.int32))[pi.clamp(min=0).long()], pi)); l_sc.append(ps)
    st_ids, st_sc = torch.stack(l_ids).contiguous(), torch.stack(l_sc).contiguous()
    h = shard.merge_topk_lists(st_ids, st_sc, K)
    assert torch.equal(h[0], ids) and torch.equal(h[1], sc)
    d_ids, d_sc = st_ids.cuda(), st_sc.cuda()
    o_ids, o_sc = torch.empty((len(qs), K), dtype=torch.int32, device="cuda"), torch.empty((len(qs), K), dtype=torch.int32, device="cuda")
    o_cnt = torch.empty((len(qs),), dtype=torch.int32, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    check(lib.s4g_merge_topk(0, 2, len(qs), K, p(d_ids), p(d_sc), K, p(o_ids), p(o_sc), p(o_cnt), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s4g_merge_topk")
    torch.cuda.synchronize()
    assert torch.equal(o_ids.cpu(), ids) and torch.equal(o_sc.cpu(), sc)
    # the RCCL path with the one rank this box has
    g_ids, g_sc, g_cnt = s4g.scan_topk(db, q, scorer, K)
    comm = shard.Comm(0, 0, 1, shard.Comm.unique_id())
    a = comm.gather_query_shards(g_ids, g_sc, g_cnt, q.n)
    torch.cuda.synchronize()
    assert torch.equal(a[0], g_ids) and torch.equal(a[1], g_sc) and torch.equal(a[2], g_cnt)
    comm.close()


# ---------------------------------------------------------------------------------------------------------------- f1 .. f4

def test_f1_device_topk_and_evalue_cut_equal_the_callback_path(s4g, sw, oracle_mod, blosum62):
    """shotgunDatabaseEValues (score cut + device top-k + exact host e-values) == steps 2-4 computed from the oracle alone,
    for caps above and below the number of significant hits and a cut nothing passes."""
    from oracle import evalue_ref as er
    L, ci, pvp = sw.L, sw.ci, sw.pvp
    rng, qs, ts = _homolog_set(691, 4, 900, every=2)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    cells = int(sum(len(t) for t in ts))
    s = sw.scorer(10, 1)
    tch, qch = [sw.chain(f"t{i}", t) for i, t in enumerate(ts)], [sw.chain(f"q{i}", q) for i, q in enumerate(qs)]
    tarr, qarr = sw.chain_array(tch), sw.chain_array(qch)
    ev = L.createEValueParams(cells, s)
    cdb = L.chainDatabaseCreate(tarr, 0, len(tch), None, 0)
    P = er.gumbel_parameters("BLOSUM_62", 10, 1, "swsharp")
    for max_al, thr in ((400, 1e-4), (12, 1e-4), (5, 1e-300)):
        b, lb = ctypes.POINTER(pvp)(), ctypes.POINTER(ci)()
        L.shotgunDatabaseEValues(ctypes.byref(b), ctypes.byref(lb), 0, qarr, len(qch), cdb, s, max_al, ev, thr, None, 0, None)
        for i in range(len(qs)):
            vals = [er.evalue(int(exp[i, t]), len(qs[i]), len(ts[t]), cells, P) for t in range(len(ts))]
            order = sorted((t for t in range(len(ts)) if vals[t] <= thr), key=lambda t: (vals[t], -int(exp[i, t]), t))[:max_al]
            got = _hits(L, sw, b[i], lb[i])
            assert [h[0] for h in got] == order and [h[1] for h in got] == [int(exp[i, t]) for t in order], (max_al, thr, i)
            assert all(h[2] == pytest.approx(vals[t], rel=1e-12, abs=0.0) for h, t in zip(got, order))
        assert (thr < 1e-100) == (sum(lb[i] for i in range(len(qs))) == 0)
        L.deleteShotgunDatabase(b, lb, len(qs))
    L.chainDatabaseDelete(cdb); L.deleteEValueParams(ev); L.scorerDelete(s)


def test_f2_traceback_of_the_survivors_equals_the_oracle(s4g, oracle_mod, blosum62, scorer):
    """the device traceback of the top-400 survivors of a scan -- one wave per pair and the 64-pairs-per-wave strip kernels,
    synchronous and as a submitted job -- start cell, end cell and the whole move string == the oracle's traceback."""
    from sift4g_amd import device as s4gd
    rng, qs, ts = _homolog_set(701, 3, 500, qlen=(40, 300), tlen=(1, 350), every=3)
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    ids, sc, cnt = [x.cpu().numpy() for x in s4g.scan_topk(db, q, scorer, 400)]
    pq = np.repeat(np.arange(len(qs)), 400)[: 3 * 400]
    keep = (np.arange(400)[None, :] < cnt[:, None]).reshape(-1)
    pq, pt, ps = pq[keep], ids.reshape(-1)[keep], sc.reshape(-1)[keep]
    assert len(pq) == 1200
    want = [oracle_mod.sw_align(qs[a], ts[b], blosum62, 10, 1) for a, b in zip(pq, pt)]
    results = {route: s4g.traceback(db, q, scorer, pq, pt, ps, route=route) for route in (None, "wave", "scan")}
    results["job"] = s4gd.TraceJob(db, q, scorer, pq, pt, ps).result()
    seen = set()
    for route, res in results.items():
        for r, o, s_ in zip(res, want, ps):
            assert o["score"] == int(s_) == r["score"]
            assert (r["q_start"], r["q_end"], r["t_start"], r["t_end"]) == (o["q_start"], o["q_end"], o["t_start"], o["t_end"]), route
            assert r["path"].tolist() == [MOVE[m] for m in o["path"]], route
            seen |= set(r["path"].tolist())
    assert seen == {1, 2, 3}


def test_f3_kmer_prefilter_equals_the_oracle_and_the_golden_lists():
    """candidate ids, counts and float scores of the device prefilter (s4g_prefilter_*) == oracle/kmer_prefilter.c bit for
    bit on a database added in two chunks, at k = 3, 4, 5 with a cut that binds; and == the committed lists of tests/golden/prefilter.json."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_prefilter_golden as g
    import oracle
    from sift4g_amd import device as s4gd

    def both(queries, chunks, k, cap):
        dev, ref = s4gd.Prefilter(queries, k, cap), oracle.Prefilter(queries, k, cap)
        for c in chunks:
            dev.add(c); ref.add(c)
        d, r = dev.finish(), ref.finish()
        dev.close()
        assert d[1].tolist() == r[1].tolist()
        for qi in range(len(queries)):
            n = int(d[1][qi])
            assert d[0][qi, :n].tolist() == r[0][qi, :n].tolist() and d[2][qi, :n].view(np.uint32).tolist() == r[2][qi, :n].view(np.uint32).tolist(), (k, qi)
        return d
    rng, qs, ts = _homolog_set(711, 12, 2500, qlen=(60, 400), tlen=(3, 600), every=9)
    for k, cap in ((5, 5000), (4, 60), (3, 60)):
        d = both(qs, [ts[:1100], ts[1100:]], k, cap)
        assert d[1].max() == (cap if cap == 60 else d[1].max()) and d[1].min() >= 1
    for c in json.load(open(os.path.join(GOLDEN, "prefilter.json"))):
        if c["nt"] > 5000:
            continue                                                         # the large cases: tests/test_gpu_prefilter.py
        cq, ct = g.make_case(c["seed"], c["nq"], c["nt"], c["repeats"])
        d = both([oracle.encode(x) for x in cq], [[oracle.encode(x) for x in ct]], c["k"], c["max_candidates"])
        got = [[int(x) for x in d[0][qi, :d[1][qi]]] for qi in range(c["nq"])]
        if c.get("digest"):
            assert g.lists_digest(got) == c["lists_sha256"], c["name"]
        else:
            for qi, want in enumerate(c["lists"]):
                assert (got[qi] == want["exact"]) if "exact" in want else (len(got[qi]) == want["size"] and set(want["must"]) <= set(got[qi])), (c["name"], qi)


def test_f4_packed_database_file_roundtrip_and_streaming(s4g, oracle_mod, blosum62, scorer, tmp_path):
    """the length-sorted packed database saved (s4g_db_save), loaded (s4g_db_load) and streamed tile by tile
    (s4g_db_open_streamed) scans like the resident one: the full matrix == the oracle, the streaming top-k identical in all three."""
    rng, qs, ts = _homolog_set(721, 4, 420, qlen=(40, 200), tlen=(1, 280), every=11)
    ts.append(rand_seqs(rng, 1, 12400, 12400)[0])                           # a sequence of the long-sequence tile class
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    db, q = s4g.Database(ts, tile_seqs=64), s4g.QuerySet(qs)
    path = tmp_path / "db.s4g"
    db.save(path)
    loaded, streamed = s4g.Database.load(path), s4g.Database.open_streamed(path)
    assert loaded.n == streamed.n == len(ts) and loaded.tiles() == streamed.tiles() == db.tiles() and len(db.tiles()) >= 7
    assert np.array_equal(s4g.scan_host(db, q, scorer), exp) and np.array_equal(s4g.scan_host(loaded, q, scorer), exp)
    want = [x.cpu().numpy() for x in s4g.scan_topk(db, q, scorer, 25, 1)]
    for other in (loaded, streamed, streamed):
        got = [x.cpu().numpy() for x in s4g.scan_topk(other, q, scorer, 25, 1)]
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    for i in range(len(qs)):
        order = sorted((j for j in range(len(ts)) if exp[i, j] >= 1), key=lambda j: (-exp[i, j], j))[:25]
        assert want[0][i].tolist() == order
    (tmp_path / "junk.s4g").write_bytes(b"not a database")
    with pytest.raises(Exception):
        s4g.Database.load(tmp_path / "junk.s4g")


# ---------------------------------------------------------------------------------------------------------------- dagger

DROPIN = os.path.join(ROOT, "build", "dropin", "sift4g")


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs /root/reference at build time: make dropin)")
def test_dagger_reference_cli_writes_the_golden_siftprediction_bytes(tmp_path):
    """north_star: "the sift4g CLI stays intact ... emitted .SIFTprediction outputs bit-exact".  The reference's UNMODIFIED sources linked
    against libswsharp / libs4g_hip (make dropin), run on its own test_files (test_files/README.md:13-19): alignments.txt and both
    .SIFTprediction files == tests/golden/dropin_predictions/, the bytes the same binary writes on the CPU oracle stand-in
    (tests/test_cpu_pipeline.py makes that comparison on the build container)."""
    tf = os.path.join(GOLDEN, "test_files")
    r = subprocess.run([DROPIN, "-q", os.path.join(tf, "query.fasta"), "--subst", tf, "-d", os.path.join(tf, "sample_protein_database.fa"),
                        "--out", str(tmp_path), "--cards", "0", "--sub-results"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    for n in ("alignments.txt", "LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction"):
        assert (tmp_path / n).read_bytes() == open(os.path.join(GOLDEN, "dropin_predictions", n), "rb").read(), n


# ---------------------------------------------------------------------------------------------------------------- cfg1 .. cfg4

def _sample_check(oracle_mod, cfg, got, pq, pt):
    exp = oracle_mod.pairs_flat(cfg["q_codes"], cfg["q_off"], cfg["t_codes"], cfg["t_off"], pq.astype(np.uint32), pt.astype(np.uint32),
                                oracle_mod.blosum62(), 10, 1, threads=os.cpu_count() or 8)
    assert np.array_equal(got[pq, pt], exp)


def test_cfg1_synthetic_queries_x_synthetic_db_reduced(s4g, oracle_mod, scorer):
    """BASELINE configs[1] (1000 queries U[300,400] x 100 000 lognormal targets, planted homologs, BLOSUM62 10/1) at 24 x 4000 with
    the same generator, on the throughput kernels the bench runs (no_pairs: at this size the library would hand part of the work to the
    latency route by itself).  The WHOLE matrix == the oracle, also on the library's own routing; every planted homolog scores above 100;
    the exact top-400 lists."""
    import torch
    from sift4g_amd import device as s4gd, synth
    cfg = synth.make_config(2, n_queries=24, n_targets=4000, homologs_per_query=20)
    db, q = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"]), s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    out = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    exp = oracle_mod.scan_flat(cfg["q_codes"], cfg["q_off"], cfg["t_codes"], cfg["t_off"], oracle_mod.blosum62(), 10, 1, threads=os.cpu_count() or 8)
    s4g.scan(db, q, scorer, out)                                              # the library's own routing for a call of this size
    assert np.array_equal(out.cpu().numpy(), exp)
    out.zero_()
    s4g.scan(db, q, scorer, out, no_pairs=True)
    st = s4gd.scan_stats(0)
    assert st["pair_items"] == 0 and st["tier_drift"] == q.n, st              # the bench's route: the drifting-frame packed kernel
    got = out.cpu().numpy()
    assert np.array_equal(got, exp)
    assert all(got[a, b] > 100 for a, b in cfg["planted"])
    ids, sc, cnt = s4g.topk(db, out, 400, 1)
    for i in range(q.n):
        order = np.lexsort((np.arange(db.n), -exp[i].astype(np.int64)))[:400]
        assert ids[i].tolist() == order.tolist() and sc[i].tolist() == exp[i, order].tolist()


def test_cfg2_swissprot_scale_generator_reduced_two_query_shards(s4g, oracle_mod, scorer):
    """BASELINE configs[2] (the Swiss-Prot-scale scan bench.py strong-scales) at 16 x 20 000 with its generator: scanned as two query
    shards (what ranks 0 and 1 of --gpus 2 run) -- the concatenation == one scan; 3000 random pairs and every planted pair == oracle."""
    import torch
    from sift4g_amd import shard, synth
    cfg = synth.make_config(3, n_queries=16, n_targets=20000, homologs_per_query=10)
    db, q = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"]), s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    whole = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, whole)
    b = shard.shard_bounds(q.n, 2)
    halves = torch.empty_like(whole)
    for r in range(2):
        s4g.scan(db, q, scorer, halves[b[r]:b[r + 1]], q_begin=b[r], q_count=b[r + 1] - b[r])
    assert torch.equal(whole, halves)
    rng = np.random.default_rng(731)
    pq = np.concatenate([rng.integers(0, q.n, 3000), [a for a, _ in cfg["planted"]]])
    pt = np.concatenate([rng.integers(0, db.n, 3000), [t for _, t in cfg["planted"]]])
    _sample_check(oracle_mod, cfg, whole.cpu().numpy(), pq, pt)


def test_cfg3_device_generated_db_shard_streaming_topk_reduced(s4g, oracle_mod, scorer):
    """BASELINE configs[3] (UniRef90 scale, DB sharded) at one shard of 60 000 device-generated targets in six tiles: streaming scan +
    running top-400 for 8 queries; a planted copy of every query leads its list with the self-score; the lists are ordered (score
    descending, id ascending); every reported hit == the oracle (residues read back from the device)."""
    import torch
    from sift4g_amd import device as s4gd, synth
    N, Q, K = 60_000, 8, 400
    dev = torch.device("cuda", 0)
    codes, offsets = synth.device_database(N, synth.SEED_BASE + 4 + 0x1000, dev)
    ql = synth.query_lengths(synth.SEED_BASE + 4 + 0x100, Q)
    q_codes, q_off = synth.random_set(synth.SEED_BASE + 4 + 0x200, ql)
    lens = (offsets[1:] - offsets[:-1]).cpu().numpy()
    planted = []
    for i in range(Q):
        p = (i + 1) * (N // (Q + 1))
        while lens[p] < ql[i]:
            p += 1
        planted.append(p)
        o = int(offsets[p])
        codes[o:o + int(ql[i])] = torch.from_numpy(q_codes[int(q_off[i]):int(q_off[i + 1])].copy()).to(dev)
    torch.cuda.synchronize()
    db = s4g.Database.from_device(codes, offsets, tile_seqs=10_000)
    assert len(db.tiles()) >= 6
    q = s4g.QuerySet(codes=q_codes, offsets=q_off)
    ids, sc, cnt = [x.cpu().numpy() for x in s4g.scan_topk(db, q, scorer, K)]
    assert (cnt == K).all()
    M = s4gd.matrix_by_name("BLOSUM_62")
    for i in range(Q):
        qc = q_codes[int(q_off[i]):int(q_off[i + 1])]
        assert ids[i, 0] == planted[i] and sc[i, 0] == int(M[qc, qc].sum())
        key = list(zip((-sc[i].astype(np.int64)).tolist(), ids[i].astype(np.int64).tolist()))
        assert key == sorted(key)
    off_h, codes_h = offsets.cpu().numpy().astype(np.uint64), codes[:int(offsets[-1])].cpu().numpy()
    pq, pt = np.repeat(np.arange(Q, dtype=np.uint32), K), ids.reshape(-1).astype(np.uint32)
    exp = oracle_mod.pairs_flat(q_codes, q_off, codes_h, off_h, pq, pt, oracle_mod.blosum62(), 10, 1, threads=os.cpu_count() or 8)
    assert np.array_equal(exp, sc.reshape(-1))


def test_cfg4_long_queries_reduced(s4g, oracle_mod, scorer):
    """BASELINE configs[4] (queries of 2000-5000 residues x the mixed-length database) at 6 x 6000 with its generator: the routing a full
    run has (drifting-frame tier by bound or speculatively, exact int32 recheck of what it marks); routing by bound alone and the
    anti-diagonal frame give the same matrix bit for bit; 2500 random pairs and every planted pair == the oracle."""
    import torch
    from sift4g_amd import device as s4gd, synth
    cfg = synth.make_config(5, n_queries=6, n_targets=6000, q_len=(2000, 5000), homologs_per_query=10)
    ql = np.diff(cfg["q_off"].astype(np.int64))
    assert ql.min() >= 2000 and ql.max() <= 5000
    db, q = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"]), s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    got = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, got, no_pairs=True)
    st = s4gd.scan_stats(0)
    assert st["tier_drift"] + st["tier_m3"] + st["tier_i16"] + st["tier_i32"] == q.n and st["pair_items"] == 0, st
    other = torch.empty_like(got)
    for mode in (dict(no_spec=True), dict(diag_frame=True), dict(force_i32=True)):
        s4g.scan(db, q, scorer, other, no_pairs=True, **mode)
        assert torch.equal(got, other), mode
    rng = np.random.default_rng(741)
    pq = np.concatenate([rng.integers(0, q.n, 2500), [a for a, _ in cfg["planted"]]])
    pt = np.concatenate([rng.integers(0, db.n, 2500), [t for _, t in cfg["planted"]]])
    _sample_check(oracle_mod, cfg, got.cpu().numpy(), pq, pt)
