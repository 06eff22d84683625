"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact int32 scores."""
import os

import numpy as np
import pytest

from conftest import ROOT, rand_seqs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def s4g():
    import sift4g_amd
    assert sift4g_amd.device_count() >= 1
    return sift4g_amd


@pytest.fixture(scope="module")
def scorer(s4g):
    return s4g.Scorer("BLOSUM_62", 10, 1)


def _check(s4g, oracle_mod, M, queries, targets, scorer, go=10, ge=1, threads=8):
    db = s4g.Database(targets)
    qs = s4g.QuerySet(queries)
    exp = oracle_mod.scan(queries, targets, M, go, ge, threads=threads)
    # default plan (fastest exact kernel per query: drifting-frame int16 / maximum3-int16 / plain packed int16 / int32),
    # without the drift kernel, the f16-arithmetic form of the maximum3 tier, no maximum3 at all, int32 alone
    # (small inputs are latency-bound: by default they take the one-wave-per-pair route, no_pairs keeps them on the scan kernels)
    # (int32 alone runs the drifting-frame int32 kernel where the scorer's shifted scores fit its int8 profile; + no_drift: the fixed frame)
    # block_pipe: the block-cooperative strip pipeline (sw_scan_lists in scan mode) wherever the queries fit its window
    # diag_frame: the drifting-frame tier in its anti-diagonal form (sw_scan_pkd, rounds 1-4) instead of the column frame (sw_scan_pkc)
    for mode in (dict(), dict(no_pairs=True), dict(diag_frame=True), dict(block_pipe=True), dict(no_drift=True), dict(f16_arith=True), dict(no_f16=True),
                 dict(force_i32=True), dict(force_i32=True, no_drift=True)):
        got = s4g.scan_host(db, qs, scorer, **mode)
        bad = np.argwhere(got != exp)
        assert bad.size == 0, (f"{mode}: {len(bad)} mismatches, first {bad[:5].tolist()} "
                               f"got {got[tuple(bad[0])]} exp {exp[tuple(bad[0])]}")
    return got


def test_micro_known_answers(s4g, oracle_mod, micro, scorer):
    qs = [oracle_mod.encode(c["q"]) for c in micro["cases"]]
    ts = [oracle_mod.encode(c["t"]) for c in micro["cases"]]
    for mode in (dict(), dict(no_pairs=True)):
        got = s4g.scan_host(s4g.Database(ts), s4g.QuerySet(qs), scorer, **mode)
        for i, c in enumerate(micro["cases"]):
            assert got[i, i] == c["score"], (mode, c["name"], int(got[i, i]), c["score"])


def test_cfg1_reference_fixture(s4g, cfg1, scorer):
    """BASELINE.json configs[0]: test_files/query.fasta x sample_protein_database.fa vs the golden 2 x 9 matrix."""
    db, q = s4g.Database([c for _, c in cfg1["targets"]]), s4g.QuerySet([c for _, c in cfg1["queries"]])
    assert np.array_equal(s4g.scan_host(db, q, scorer), cfg1["scores"])                    # latency route (pairs)
    assert np.array_equal(s4g.scan_host(db, q, scorer, no_pairs=True), cfg1["scores"])     # scan kernels


def test_random_standard_alphabet(s4g, oracle_mod, blosum62, scorer):
    rng = np.random.default_rng(11)
    _check(s4g, oracle_mod, blosum62, rand_seqs(rng, 24, 1, 450), rand_seqs(rng, 700, 1, 600), scorer)


def test_all_26_codes_and_ragged_groups(s4g, oracle_mod, blosum62, scorer):
    """every letter code (B, J, O, U, X, Z included); target count not a multiple of 64."""
    rng = np.random.default_rng(12)
    _check(s4g, oracle_mod, blosum62, rand_seqs(rng, 9, 1, 200, 26), rand_seqs(rng, 131, 1, 300, 26), scorer)


def test_strip_and_chunk_boundaries(s4g, oracle_mod, blosum62, scorer):
    """query lengths around the strip height (32) and target lengths around the 4-column chunk."""
    rng = np.random.default_rng(13)
    qlens = [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 95, 96, 97, 128, 129]
    tlens = [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 127, 128, 129, 255, 256, 257]
    qs = [rand_seqs(rng, 1, l, l)[0] for l in qlens]
    ts = [rand_seqs(rng, 1, l, l)[0] for l in tlens]
    _check(s4g, oracle_mod, blosum62, qs, ts, scorer)


def test_empty_and_degenerate_inputs(s4g, oracle_mod, blosum62, scorer):
    rng = np.random.default_rng(14)
    e = np.zeros(0, dtype=np.uint8)
    qs = [e, rand_seqs(rng, 1, 40, 40)[0], e, rand_seqs(rng, 1, 1, 1)[0]]
    ts = [rand_seqs(rng, 1, 50, 50)[0], e, e, rand_seqs(rng, 1, 1, 1)[0], rand_seqs(rng, 1, 70, 70)[0]]
    got = _check(s4g, oracle_mod, blosum62, qs, ts, scorer)
    assert (got[0] == 0).all() and (got[:, 1] == 0).all()
    # zero queries / zero targets: calls succeed and return empty results
    assert s4g.scan_host(s4g.Database(ts), s4g.QuerySet([]), scorer).shape == (0, 5)
    assert s4g.scan_host(s4g.Database([]), s4g.QuerySet(qs), scorer).shape == (4, 0)


def test_homologs_identical_and_gapped(s4g, oracle_mod, blosum62, scorer):
    """high-scoring pairs: identical copies, planted indels of 1..40 residues, long runs of W (max score/column)."""
    rng = np.random.default_rng(15)
    qs = rand_seqs(rng, 6, 200, 420)
    ts = []
    for q in qs:
        ts.append(q.copy())
        for gap in (1, 2, 3, 10, 40):
            cut = int(rng.integers(20, len(q) - 20))
            ts.append(np.concatenate([q[:cut], rand_seqs(rng, 1, gap, gap)[0], q[cut:]]))   # insertion in target
            ts.append(np.concatenate([q[:cut], q[cut + gap:]]))                             # deletion in target
    w = np.full(500, ord("W") - 65, dtype=np.uint8)
    qs.append(w[:400]); ts.append(w)
    got = _check(s4g, oracle_mod, blosum62, qs, ts, scorer)
    assert got[-1, -1] == 400 * 11


def test_long_sequences(s4g, oracle_mod, blosum62, scorer):
    """BASELINE config 5 shape at oracle-friendly size: queries 2000-5000, targets up to 3000 (+ one 12k)."""
    rng = np.random.default_rng(16)
    qs = rand_seqs(rng, 3, 2000, 5000)
    ts = rand_seqs(rng, 70, 30, 3000) + [rand_seqs(rng, 1, 12000, 12000)[0], np.concatenate([qs[0][100:1900], qs[1][:50]])]
    _check(s4g, oracle_mod, blosum62, qs, ts, scorer)


def test_giant_queries_and_targets(s4g, oracle_mod, blosum62, scorer):
    """titin-sized QUERIES (36 000 and 20 000 residues: score bounds of 200 000 and 110 000, beyond every int16 window) against a small
    database that holds a copy of the first (self-score 209 182), a 14 500-residue piece of the second and an unrelated 35 000-residue
    target: the library's own routing for the call, the streaming top-k, both pair-list routes, and the device traceback of the
    36 000 x 36 000 self-alignment -- scores, lists, coordinates and path lengths == the oracle."""
    rng = np.random.default_rng(77)
    qs = [rand_seqs(rng, 1, 36000, 36000)[0], rand_seqs(rng, 1, 20000, 20000)[0], rand_seqs(rng, 1, 350, 350)[0]]
    ts = rand_seqs(rng, 100, 30, 900) + [qs[0].copy(), rand_seqs(rng, 1, 35000, 35000)[0], qs[1][500:15000].copy()]
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=os.cpu_count() or 8)
    assert exp[0, 100] > 200000 and exp[1, 102] > 80000
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    assert np.array_equal(s4g.scan_host(db, q, scorer), exp)
    ids, sc, cnt = [x.cpu().numpy() for x in s4g.scan_topk(db, q, scorer, 5, 1)]
    for i in range(len(qs)):
        order = sorted(range(len(ts)), key=lambda j: (-exp[i, j], j))[:5]
        assert ids[i].tolist() == order and sc[i].tolist() == [int(exp[i, j]) for j in order]
    pq, pt = np.array([0, 1, 2, 0, 1], dtype=np.uint32), np.array([100, 102, 5, 101, 101], dtype=np.uint32)
    for route in ("lists", "wave"):
        assert s4g.score_pairs(db, q, scorer, pq, pt, route=route).tolist() == exp[pq, pt].tolist(), route
    res = s4g.traceback(db, q, scorer, pq[:3], pt[:3], exp[pq[:3], pt[:3]])
    for r, a, b in zip(res, pq[:3], pt[:3]):
        o = oracle_mod.sw_align(qs[a], ts[b], blosum62, 10, 1)
        assert (r["q_start"], r["q_end"], r["t_start"], r["t_end"]) == (o["q_start"], o["q_end"], o["t_start"], o["t_end"])
        assert r["path"].tolist() == [{0: 1, 1: 2, 2: 3}[m] for m in o["path"]]


def test_more_items_than_resident_waves(s4g, oracle_mod, blosum62, scorer):
    """enough (group, query) items that the persistent grid loops, with very unequal work."""
    rng = np.random.default_rng(17)
    qs = rand_seqs(rng, 40, 5, 120)
    ts = rand_seqs(rng, 64 * 40 + 5, 1, 90)
    _check(s4g, oracle_mod, blosum62, qs, ts, scorer)


@pytest.mark.parametrize("go,ge", [(11, 1), (10, 2), (5, 5), (12, 0), (1, 1)])
def test_other_gap_penalties(s4g, oracle_mod, blosum62, go, ge):
    rng = np.random.default_rng(18 + go)
    sc = s4g.Scorer("BLOSUM_62", go, ge)
    qs = rand_seqs(rng, 6, 30, 200)
    ts = rand_seqs(rng, 100, 10, 250) + [np.concatenate([qs[0][:60], qs[0][70:]])]
    _check(s4g, oracle_mod, blosum62, qs, ts, sc, go, ge)


def test_int16_bound_routing(s4g, oracle_mod, blosum62, scorer):
    """queries whose score bound exceeds int16 must leave the packed kernel: W-rich long queries (11 per residue)
    mixed with ordinary ones, against targets that actually reach > 32767."""
    rng = np.random.default_rng(25)
    W = ord("W") - 65
    long_w = np.full(3300, W, dtype=np.uint8)                 # bound 36300 > 32767 -> int32 kernel
    edge_w = np.full(2978, W, dtype=np.uint8)                 # bound 32758 <= 32767 -> packed kernel, score at the edge
    qs = [long_w, edge_w] + rand_seqs(rng, 5, 100, 400)
    ts = [np.full(3400, W, dtype=np.uint8), np.full(2978, W, dtype=np.uint8)] + rand_seqs(rng, 70, 50, 500)
    got = _check(s4g, oracle_mod, blosum62, qs, ts, scorer)
    assert got[0, 0] == 3300 * 11 and got[1, 0] == 2978 * 11 == 32758 and got[1, 1] == 32758


def test_f16_window_routing(s4g, oracle_mod, blosum62, scorer):
    """the packed f16 kernel is exact on scores up to 4096 (integers H - 2048 in f16): queries whose bound is just
    inside / outside the window, scores that reach the top of the window, and large gap penalties."""
    W, A = ord("W") - 65, 0
    inside = np.full(372, W, dtype=np.uint8)                   # bound 4092 <= 4096 -> f16 kernel, score up to 4092
    outside = np.full(373, W, dtype=np.uint8)                  # bound 4103 -> packed int16
    rng = np.random.default_rng(32)
    qs = [inside, outside, np.full(1000, A, dtype=np.uint8)] + rand_seqs(rng, 5, 100, 700)   # poly-A 1000: bound 4000
    ts = [np.full(400, W, dtype=np.uint8), np.full(372, W, dtype=np.uint8), np.full(1500, A, dtype=np.uint8),
          np.concatenate([np.full(186, W, dtype=np.uint8), rand_seqs(rng, 1, 3, 3)[0], np.full(186, W, dtype=np.uint8)])]
    ts += rand_seqs(rng, 80, 50, 600)
    got = _check(s4g, oracle_mod, blosum62, qs, ts, scorer)
    assert got[0, 0] == 4092 and got[1, 0] == 4103 and got[2, 2] == 4000
    for go, ge in ((2048, 2048), (40, 1), (3000, 7)):          # gap penalties at / beyond the f16 window
        sc = s4g.Scorer("BLOSUM_62", go, ge)
        _check(s4g, oracle_mod, blosum62, qs[:4], ts[:30], sc, go, ge)


def test_maximum3_int16_window_routing(s4g, oracle_mod, blosum62, scorer):
    """ArithI16M3 (v_pk_maximum3_f16 on biased int16 bit patterns) is exact while bound + bias <= 0x7BFF = 31743:
    scores right at the top of the window, small values (f16 denormal patterns), large gap penalties (big bias)."""
    W = ord("W") - 65
    rng = np.random.default_rng(33)
    top = np.full(2879, W, dtype=np.uint8)                     # bound 31669 + bias 64 = 31733 <= 31743 -> maximum3 tier
    over = np.full(2880, W, dtype=np.uint8)                    # bound 31680 + 64 > 31743 -> plain packed int16
    qs = [top, over] + rand_seqs(rng, 4, 5, 60)                # short queries: every score stays in the denormal range
    ts = [np.full(3000, W, dtype=np.uint8), np.full(2879, W, dtype=np.uint8)] + rand_seqs(rng, 60, 1, 80)
    got = _check(s4g, oracle_mod, blosum62, qs, ts, scorer)
    assert got[0, 0] == 31669 and got[1, 0] == 31680
    for go, ge in ((5000, 3), (31000, 31000)):                 # bias = open: shrinks the window, far beyond any real penalty
        sc = s4g.Scorer("BLOSUM_62", go, ge)
        _check(s4g, oracle_mod, blosum62, qs[2:] + [np.full(60, W, dtype=np.uint8)], ts[1:40], sc, go, ge)


def test_drift_frame_window_routing(s4g, oracle_mod, blosum62):
    """sw_scan_pkd stores X + B + (i+j)*ext: exact while bound + B + (rows + columns)*ext <= 0x7BFF.  Large ext makes the
    drift dominate; long targets push single tiles out of the window while short-target tiles stay in."""
    W = ord("W") - 65
    rng = np.random.default_rng(34)
    qs = rand_seqs(rng, 6, 50, 400) + [np.full(1500, W, dtype=np.uint8)]
    ts = rand_seqs(rng, 150, 10, 500) + [rand_seqs(rng, 1, 13000, 13000)[0], np.full(2000, W, dtype=np.uint8),
                                          np.concatenate([qs[0], qs[0], qs[0]])]
    # (the last two: frames of (rows + columns) * ext = 10^7 .. 6 * 10^7 in the int32 drifting kernel, whose guard against leaving
    # int32 -- half a million residues at ext 4096 -- lets them through)
    for go, ge in ((10, 1), (12, 4), (30, 30), (11, 0), (200, 9), (3000, 2000), (4096, 4096)):
        sc = s4g.Scorer("BLOSUM_62", go, ge)
        got = _check(s4g, oracle_mod, blosum62, qs, ts, sc, go, ge)
        assert got[6, 151] == 1500 * 11
    db = s4g.Database(ts, tile_seqs=64)                        # the 13000-residue target gets a tile of its own
    exp = oracle_mod.scan(qs, ts, blosum62, 12, 3, threads=8)
    assert np.array_equal(s4g.scan_host(db, s4g.QuerySet(qs), s4g.Scorer("BLOSUM_62", 12, 3)), exp)


def test_column_frame_window_edge(s4g, oracle_mod, blosum62, scorer):
    """sw_scan_pkc stores X + B + j*ext and keeps its running maximum in that frame: exact while bound + B + columns*ext <= 0x7BFF,
    whatever the query's length adds in rows.  A poly-W query of 2600 against a poly-W target of 3068: bound 28 600 + 64 + 3076
    columns = 31 740 <= 31 743 routes by bound into the drift tier and the in-frame maximum really climbs to 31 732, twelve below the
    first f16 infinity pattern; one more W in the query (28 611) is beyond the window and runs elsewhere (speculatively in the
    same kernel by default -- marked and rechecked --, in the maximum3 tier with routing by bound).  Both frames, every mode."""
    from sift4g_amd import device as s4gd
    W = ord("W") - 65
    rng = np.random.default_rng(35)
    qs = [np.full(2600, W, dtype=np.uint8), np.full(2601, W, dtype=np.uint8)] + rand_seqs(rng, 4, 100, 400)
    ts = [np.full(3068, W, dtype=np.uint8), np.full(2600, W, dtype=np.uint8)] + rand_seqs(rng, 100, 30, 700)
    got = _check(s4g, oracle_mod, blosum62, qs, ts, scorer)
    assert got[0, 0] == 28600 and got[1, 0] == 28611 and got[1, 1] == 28600
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    # (a tier with little work is folded into the next slower one when one launch beats two, so the routing is read off calls
    # whose queries all belong to one tier)
    inside = s4g.QuerySet([qs[0]] + qs[2:])
    assert np.array_equal(s4g.scan_host(db, inside, scorer, no_pairs=True, no_spec=True), exp[[0, 2, 3, 4, 5]])
    st = s4gd.scan_stats(0)
    assert st["tier_drift"] == 5 and st["tier_m3"] == 0 and st["speculative"] == 0 and st["rechecked"] == 0, st
    assert np.array_equal(s4g.scan_host(db, s4g.QuerySet([qs[1]]), scorer, no_pairs=True, no_spec=True), exp[1:2])
    st = s4gd.scan_stats(0)
    assert st["tier_drift"] == 0 and st["tier_m3"] == 1, st
    assert np.array_equal(s4g.scan_host(db, q, scorer, no_pairs=True), exp)
    st = s4gd.scan_stats(0)
    assert st["tier_drift"] == 6 and st["speculative"] == 1 and 1 <= st["rechecked"] <= 4, st
    # the anti-diagonal frame's window also counts the rows: the 2600-residue query is outside it (28 600 + 64 + 2664 + 3076)
    assert np.array_equal(s4g.scan_host(db, s4g.QuerySet([qs[0]]), scorer, diag_frame=True, no_spec=True), exp[0:1])
    st = s4gd.scan_stats(0)
    assert st["tier_drift"] == 0 and st["tier_m3"] == 1, st


def test_odd_query_counts_and_unequal_pairs(s4g, oracle_mod, blosum62, scorer):
    """packed kernel pairs queries of similar length: odd counts, a lone query, very unequal partners."""
    rng = np.random.default_rng(26)
    ts = rand_seqs(rng, 200, 5, 300)
    for lens in ([77], [10, 700], [33, 34, 35], [1, 2, 3, 250, 251], [64, 64, 64, 64, 64, 64, 65]):
        qs = [rand_seqs(rng, 1, l, l)[0] for l in lens]
        _check(s4g, oracle_mod, blosum62, qs, ts, scorer)


def test_custom_matrix(s4g, oracle_mod):
    """scorer tables other than BLOSUM62: random symmetric table with large entries."""
    rng = np.random.default_rng(19)
    M = rng.integers(-20, 21, size=(26, 26)).astype(np.int32)
    M = np.triu(M) + np.triu(M, 1).T
    sc = s4g.Scorer(M, 7, 2)
    _check(s4g, oracle_mod, M, rand_seqs(rng, 5, 20, 150, 26), rand_seqs(rng, 90, 5, 200, 26), sc, 7, 2)


@pytest.mark.parametrize("name,go,ge", [("BLOSUM_45", 14, 2), ("BLOSUM_50", 13, 2), ("BLOSUM_80", 10, 1), ("BLOSUM_90", 10, 1),
                                        ("PAM_30", 9, 1), ("PAM_70", 10, 1), ("PAM_250", 14, 2), ("BLOSUM_250", 10, 1)])
def test_every_builtin_matrix(s4g, oracle_mod, name, go, ge):
    """the other tables scorerCreateMatrix knows (main.cpp:304-314), each with BLAST's default penalties for it: every tier
    against the oracle on all 26 codes (B / Z / X columns, J / O / U as X), with planted homologs.  PAM_30's -17 .. 13 range
    and BLOSUM_45's 15 move the bias and the score bounds the tier routing works from."""
    from sift4g_amd.device import matrix_by_name
    M = matrix_by_name(name)
    rng = np.random.default_rng(sum(map(ord, name)))
    qs = rand_seqs(rng, 11, 40, 420, 26)
    ts = rand_seqs(rng, 520, 1, 500, 26)
    for k in range(40):
        src = qs[k % len(qs)]
        m = src.copy()
        idx = rng.choice(len(m), size=max(1, len(m) // 5), replace=False)
        m[idx] = rand_seqs(rng, 1, len(idx), len(idx))[0]
        cut = int(rng.integers(1, len(m)))
        ts[k * 7] = np.concatenate([m[:cut], rand_seqs(rng, 1, 3, 3)[0], m[cut:]])
    got = _check(s4g, oracle_mod, M, qs, ts, s4g.Scorer(name, go, ge), go, ge)
    assert got.max() > 300


def test_query_ranges_and_device_tensor(s4g, oracle_mod, blosum62, scorer):
    """s4g_scan into a torch device buffer on torch's current stream, by query sub-range."""
    import torch
    rng = np.random.default_rng(20)
    qs, ts = rand_seqs(rng, 10, 20, 150), rand_seqs(rng, 200, 5, 180)
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    out = torch.full((4, len(ts)), -7, dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, out, q_begin=3, q_count=4)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), exp[3:7])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        out2 = torch.empty((10, len(ts)), dtype=torch.int32, device="cuda:0")
        s4g.scan(db, q, scorer, out2)
    side.synchronize()
    assert np.array_equal(out2.cpu().numpy(), exp)


def test_candidate_lists(s4g, oracle_mod, blosum62, scorer):
    """the per-query candidate lists sift4g's wrapper builds (database_alignment.cpp:136-162)."""
    rng = np.random.default_rng(21)
    qs, ts = rand_seqs(rng, 7, 20, 150), rand_seqs(rng, 300, 5, 180)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    lists = [np.sort(rng.choice(300, size=int(rng.integers(0, 60)), replace=False)).astype(np.uint32) for _ in qs]
    off = np.zeros(len(qs) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(l) for l in lists])
    ids = np.concatenate(lists) if off[-1] else np.zeros(0, np.uint32)
    got = s4g.score_lists(s4g.Database(ts), s4g.QuerySet(qs), scorer, off, ids)
    want = np.concatenate([exp[i, l] for i, l in enumerate(lists)])
    assert np.array_equal(got, want)


def test_warmup_then_scan_in_a_fresh_process():
    """s4g_warmup (what cudaCheckCards starts, main.cpp:186): a fresh process warms the device on a thread of its own and scans
    right away -- the scan waits for the warm-up on the device mutex and gives the oracle's scores; calling it again is a no-op."""
    import subprocess, sys
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from sift4g_amd import _lib
assert _lib.lib.s4g_warmup(0) == 0 and _lib.lib.s4g_warmup(0) == 0
import numpy as np, oracle, sift4g_amd as s4g
from conftest import rand_seqs
rng = np.random.default_rng(5)
qs, ts = rand_seqs(rng, 3, 50, 300), rand_seqs(rng, 200, 1, 400)
got = s4g.scan_host(s4g.Database(ts), s4g.QuerySet(qs), s4g.Scorer("BLOSUM_62", 10, 1))
assert np.array_equal(got, oracle.scan(qs, ts, oracle.blosum62(), 10, 1, threads=4))
pq = np.repeat(np.arange(3, dtype=np.uint32), 200); pt = np.tile(np.arange(200, dtype=np.uint32), 3)
assert np.array_equal(s4g.score_pairs(s4g.Database(ts), s4g.QuerySet(qs), s4g.Scorer("BLOSUM_62", 10, 1), pq, pt, route="lists"), got.reshape(-1))
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-2000:])


def test_candidate_lists_on_the_packed_list_kernel(s4g, oracle_mod, blosum62):
    """route "lists" of s4g_score_pairs_host (sw_scan_lists: a query's candidates sorted by length on the device, 128 to an item,
    two targets per register) against the oracle: list sizes around the item size (1, 127, 128, 129, 300), empty targets and an
    empty query, all 26 codes, a 2900-residue target, queries up to 2600 residues (the frame window decides per query what may
    go to the kernel; the rest runs as pair waves in the same call), several scorers incl. extend 0 and one whose penalties
    leave no window at all."""
    from sift4g_amd.device import matrix_by_name
    rng = np.random.default_rng(91)
    qs = rand_seqs(rng, 9, 30, 420, 26) + [np.zeros(0, dtype=np.uint8), rand_seqs(rng, 1, 2600, 2600)[0], rand_seqs(rng, 1, 1100, 1100)[0]]
    ts = rand_seqs(rng, 700, 1, 520, 26) + [np.zeros(0, dtype=np.uint8)] * 3 + rand_seqs(rng, 6, 900, 2900, 26)
    for k in range(60):                                            # homologs with an indel
        src = qs[k % 9]
        m = src.copy()
        idx = rng.choice(len(m), size=max(1, len(m) // 6), replace=False)
        m[idx] = rand_seqs(rng, 1, len(idx), len(idx))[0]
        cut = int(rng.integers(1, len(m)))
        ts[k * 11] = np.concatenate([m[:cut], rand_seqs(rng, 1, 4, 4)[0], m[cut:]])
    ts[5] = qs[10][300:2300].copy(); ts[6] = qs[11].copy()
    sizes = [1, 127, 128, 129, 300, 64, 709, 0, 2, 5, 400, 260]
    pq, pt = [], []
    for i, n in enumerate(sizes):
        cand = rng.choice(len(ts), size=n, replace=False) if n else np.zeros(0, dtype=np.int64)
        pq += [i] * n; pt += cand.tolist()
    pq += [10, 11, 0]; pt += [5, 6, len(ts) - 1]
    pq, pt = np.array(pq, dtype=np.uint32), np.array(pt, dtype=np.uint32)
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    for name, go, ge in (("BLOSUM_62", 10, 1), ("BLOSUM_62", 7, 0), ("PAM_30", 9, 1), ("BLOSUM_45", 14, 2), ("BLOSUM_62", 40, 25), ("BLOSUM_80", 5, 5)):
        M = matrix_by_name(name)
        exp = oracle_mod.pairs_flat(*oracle_mod._flat(qs), *oracle_mod._flat(ts), pq, pt, M, go, ge, threads=8)
        sc = s4g.Scorer(name, go, ge)
        for route in ("lists", "wave", None):
            got = s4g.score_pairs(db, q, sc, pq, pt, route=route)
            bad = np.nonzero(got != exp)[0]
            assert bad.size == 0, (name, go, ge, route, len(bad), [(int(pq[b]), int(pt[b]), int(got[b]), int(exp[b])) for b in bad[:5]])
    assert exp.max() > 100


def test_candidate_lists_large_batch_takes_the_list_route_by_itself(s4g, scorer):
    """the batched wrapper's shape at a third of its size: 240 000 pairs in one call, enough for the library's estimate to send the
    lists to sw_scan_lists on its own (a launch of lists cannot end before its longest item, so small batches stay with the pair
    waves, and so do this batch's longest targets); every score equals the full scan's."""
    import torch
    from sift4g_amd import synth
    from sift4g_amd import device as s4gd
    c = synth.make_config(2, 120, 20000)
    db = s4g.Database(codes=c["t_codes"], offsets=c["t_off"])
    q = s4g.QuerySet(codes=c["q_codes"], offsets=c["q_off"])
    rng = np.random.default_rng(3)
    pt = np.concatenate([np.sort(rng.choice(db.n, size=2000, replace=False)) for _ in range(q.n)]).astype(np.uint32)
    pq = np.repeat(np.arange(q.n, dtype=np.uint32), 2000)
    full = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, full, no_pairs=True)
    want = full.cpu().numpy()[pq, pt]
    got = s4g.score_pairs(db, q, scorer, pq, pt)
    st = s4gd.scan_stats(0)
    assert np.array_equal(got, want)
    assert st["list_pairs"] > len(pq) // 2 and st["pair_items"] > 0 and st["list_pairs"] + st["pair_items"] == len(pq), st
    assert np.array_equal(s4g.score_pairs(db, q, scorer, pq, pt, route="lists"), want)


def test_topk_matches_sorted_oracle(s4g, oracle_mod, blosum62, scorer):
    import torch
    rng = np.random.default_rng(22)
    qs, ts = rand_seqs(rng, 6, 50, 120), rand_seqs(rng, 1500, 20, 150)
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    out = torch.empty((len(qs), len(ts)), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, out)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    for k, min_score in ((10, 1), (400, 30), (2000, 1)):
        ids, sc, cnt = s4g.topk(db, out, k, min_score)
        for i in range(len(qs)):
            order = sorted((j for j in range(len(ts)) if exp[i, j] >= min_score), key=lambda j: (-exp[i, j], j))[:k]
            assert cnt[i] == len(order)
            assert ids[i, :cnt[i]].tolist() == order
            assert sc[i, :cnt[i]].tolist() == [int(exp[i, j]) for j in order]
        dids, dsc, dcnt = s4g.topk_device(db, out, k, min_score)                # device-resident form (feeds the RCCL gather)
        torch.cuda.synchronize()
        assert np.array_equal(dcnt.cpu().numpy().astype(np.uint32), cnt)
        assert np.array_equal(dsc.cpu().numpy(), sc)
        assert np.array_equal(dids.cpu().numpy().astype(np.uint32), ids)


def test_topk_heavy_ties_and_large_scores(s4g, scorer):
    """exactness of the radix select: thousands of ties at the cut, scores beyond 12 bits, fewer than k candidates."""
    import torch
    rng = np.random.default_rng(24)
    n = 5000
    db = s4g.Database([np.zeros(3, dtype=np.uint8)] * n)
    rows = np.stack([
        np.full(n, 37),                                                    # all tied
        rng.integers(30, 34, size=n),                                      # 4 distinct values
        np.where(rng.random(n) < 0.01, rng.integers(4000, 300000, size=n), rng.integers(0, 60, size=n)),   # > 12-bit scores
        np.where(np.arange(n) % 1000 == 7, 55, 0),                         # only 5 candidates
        np.zeros(n),                                                       # none
        # beyond 24 bits (round 1 clamped these silently): ties in the upper 24 bits that only the low bits order, up to INT32_MAX
        np.where(rng.random(n) < 0.3, (1 << 30) + rng.integers(0, 128, size=n), rng.integers(0, 1 << 24, size=n)),
        np.where(rng.random(n) < 0.5, (1 << 25) + rng.integers(0, 4, size=n), (1 << 25) - rng.integers(1, 3, size=n)),
        np.where(np.arange(n) == 4321, 2**31 - 1, rng.integers(0, 2**31 - 1, size=n)),
    ]).astype(np.int32)
    t = torch.from_numpy(rows).cuda()
    for k in (1, 7, 400, 4096):
        ids, sc, cnt = s4g.topk(db, t, k, 1)
        for i in range(rows.shape[0]):
            order = sorted((j for j in range(n) if rows[i, j] >= 1), key=lambda j: (-rows[i, j], j))[:k]
            assert cnt[i] == len(order) and ids[i, :cnt[i]].tolist() == order
            assert sc[i, :cnt[i]].tolist() == [int(rows[i, j]) for j in order]
            assert (ids[i, cnt[i]:] == 0xFFFFFFFF).all() and (sc[i, cnt[i]:] == 0).all()


def test_topk_fast_path_edges(s4g, scorer):
    """topk_rows' fast path (round 5: a per-thread histogram of 63 bins of four scores, then the scores at or above the cut's bin
    collected and sorted) and the general path behind it, on rows built to sit on both sides of every branch: a protein-like
    distribution with a few large hits (fast path); the k-th score exactly on a bin edge; more than 8192 scores in the cut's bin
    (general path); the cut beyond 251 (general path); fewer than k scores above min_score; min_score inside the cut's bin; row
    lengths that are not multiples of 4 (the rows of a matrix then start at every alignment), thousands of ties at the cut."""
    import torch
    rng = np.random.default_rng(25)
    for n in (9001, 9002, 9003, 20000):
        db = s4g.Database([np.zeros(3, dtype=np.uint8)] * n)
        prot = np.clip(rng.normal(45, 8, size=n), 0, None).astype(np.int64)
        prot[rng.choice(n, 40, replace=False)] = rng.integers(300, 2500, size=40)
        edge = np.where(np.arange(n) < 400, 64, rng.integers(0, 64, size=n))                    # exactly 400 scores >= 64 = a bin's lower edge
        crowded = rng.integers(60, 64, size=n)                                                   # every score in one bin: > 8192 there
        high = rng.integers(200, 900, size=n)                                                    # the 400th largest lies beyond 251
        few = np.where(rng.random(n) < 0.01, rng.integers(1, 50, size=n), 0)
        ties = np.where(rng.random(n) < 0.5, 66, rng.integers(0, 66, size=n))                    # thousands of ties at the cut
        ramp = np.arange(n) % 240
        rows = np.stack([prot, edge, crowded, high, few, ties, ramp, np.zeros(n, dtype=np.int64)]).astype(np.int32)
        t = torch.from_numpy(rows).cuda()
        for k, ms in ((1, 1), (37, 1), (400, 1), (400, 65), (400, 67), (2000, 30), (4096, 1)):
            ids, sc, cnt = s4g.topk(db, t, k, ms)
            for i in range(rows.shape[0]):
                order = np.lexsort((np.arange(n), -rows[i].astype(np.int64)))
                order = order[rows[i][order] >= ms][:k]
                assert cnt[i] == len(order), (n, k, ms, i)
                assert np.array_equal(ids[i, :cnt[i]].astype(np.int64), order), (n, k, ms, i)
                assert np.array_equal(sc[i, :cnt[i]], rows[i][order]), (n, k, ms, i)
                assert (ids[i, cnt[i]:] == 0xFFFFFFFF).all() and (sc[i, cnt[i]:] == 0).all()


def test_tiled_database_scan(s4g, oracle_mod, blosum62, scorer):
    """the database is stored as tiles of the length-sorted order (incl. a separate long-sequence class):
    full-matrix scans must not depend on the tiling."""
    rng = np.random.default_rng(27)
    qs = rand_seqs(rng, 9, 20, 260)
    ts = rand_seqs(rng, 700, 1, 300) + [rand_seqs(rng, 1, 12300, 12300)[0], rand_seqs(rng, 1, 13000, 13000)[0]]
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    q = s4g.QuerySet(qs)
    for tile_seqs in (0, 64, 192, 1000):
        db = s4g.Database(ts, tile_seqs=tile_seqs)
        tiles = db.tiles()
        assert sum(tiles) == len(ts)
        if tile_seqs == 64:
            assert len(tiles) >= 11 and tiles[0] == 2          # the two > 12288-residue sequences form their own tile (side stream)
        # (small inputs are latency-bound: by default they take the one-wave-per-pair route, no_pairs keeps them on the scan kernels)
    # (int32 alone runs the drifting-frame int32 kernel where the scorer's shifted scores fit its int8 profile; + no_drift: the fixed frame)
    # block_pipe: the block-cooperative strip pipeline (sw_scan_lists in scan mode) wherever the queries fit its window
    # diag_frame: the drifting-frame tier in its anti-diagonal form (sw_scan_pkd, rounds 1-4) instead of the column frame (sw_scan_pkc)
    for mode in (dict(), dict(no_pairs=True), dict(diag_frame=True), dict(block_pipe=True), dict(no_drift=True), dict(f16_arith=True), dict(no_f16=True),
                 dict(force_i32=True), dict(force_i32=True, no_drift=True)):
            assert np.array_equal(s4g.scan_host(db, q, scorer, **mode), exp)


def test_streaming_scan_topk_equals_full_matrix_topk(s4g, oracle_mod, blosum62, scorer):
    """s4g_scan_topk (tile by tile, running merge, score matrix never materialised) == s4g_topk of the full scan,
    including ties that straddle tiles."""
    import torch
    rng = np.random.default_rng(28)
    qs = rand_seqs(rng, 11, 30, 200)
    ts = rand_seqs(rng, 900, 5, 260)
    for j in range(0, 900, 13):                       # many equal-score targets (identical copies) spread over all tiles
        ts[j] = qs[j % 11][:60].copy()
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    q = s4g.QuerySet(qs)
    for tile_seqs in (0, 128, 320):
        db = s4g.Database(ts, tile_seqs=tile_seqs)
        full = torch.empty((len(qs), len(ts)), dtype=torch.int32, device="cuda:0")
        s4g.scan(db, q, scorer, full)
        for k, ms in ((1, 1), (25, 1), (400, 20), (1500, 1)):
            want = s4g.topk(db, full, k, ms)
            # scan kernels only: with three or more tiles consecutive tiles alternate between two streams / buffers / workspaces
            alt = s4g.scan_topk(db, q, scorer, k, ms, no_pairs=True)
            ids, sc, cnt = s4g.scan_topk(db, q, scorer, k, ms)
            torch.cuda.synchronize()
            assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), want[2])
            assert np.array_equal(sc.cpu().numpy(), want[1])
            assert np.array_equal(ids.cpu().numpy().astype(np.uint32), want[0])
            assert all(np.array_equal(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(alt, (ids, sc, cnt))), (tile_seqs, k)
            for i in range(len(qs)):                   # and against the oracle directly
                order = sorted((j for j in range(len(ts)) if exp[i, j] >= ms), key=lambda j: (-exp[i, j], j))[:k]
                assert ids[i, :len(order)].cpu().tolist() == order
        sub = s4g.scan_topk(db, q, scorer, 10, 1, q_begin=4, q_count=3)
        assert np.array_equal(sub[0].cpu().numpy(), s4g.scan_topk(db, q, scorer, 10, 1)[0][4:7].cpu().numpy())


def test_streaming_topk_segments_for_small_batches(s4g, scorer):
    """few queries x a wide tile: the tile's top-k runs as several column segments per row whose lists are merged.  70 000
    short targets (scores in a narrow range: thousands of ties at every cut, across segment borders), 1 .. 5 queries,
    k from 1 to 1500, also tiled -- the same lists as the top-k of the full score matrix."""
    import torch
    rng = np.random.default_rng(71)
    qs = rand_seqs(rng, 5, 40, 90)
    ts = rand_seqs(rng, 70000, 8, 40)
    for j in range(0, 70000, 997):
        ts[j] = qs[j % 5][:30].copy()
    q = s4g.QuerySet(qs)
    for tile_seqs in (0, 40000, 16384):              # 16384: five tiles, scan kernels only -> alternating tiles
        db = s4g.Database(ts, tile_seqs=tile_seqs)
        full = torch.empty((len(qs), len(ts)), dtype=torch.int32, device="cuda:0")
        s4g.scan(db, q, scorer, full)
        for nq in (1, 2, 5):
            for k, ms in ((1, 1), (400, 1), (1500, 12)):
                want = s4g.topk(db, full[:nq].contiguous(), k, ms)
                ids, sc, cnt = s4g.scan_topk(db, q, scorer, k, ms, q_begin=0, q_count=nq, no_pairs=(tile_seqs == 16384))
                torch.cuda.synchronize()
                assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), want[2]), (tile_seqs, nq, k)
                assert np.array_equal(sc.cpu().numpy(), want[1]), (tile_seqs, nq, k)
                assert np.array_equal(ids.cpu().numpy().astype(np.uint32), want[0]), (tile_seqs, nq, k)
        row = full[0].cpu().numpy()                      # and one row against numpy directly
        order = sorted(range(len(ts)), key=lambda j: (-int(row[j]), j))[:400]
        got = s4g.scan_topk(db, q, scorer, 400, 1, q_begin=0, q_count=1)[0][0].cpu().tolist()
        assert got == order


def test_device_traceback_equals_oracle(s4g, oracle_mod, blosum62, scorer):
    """s4g_traceback_host: end cell, start cell and the whole move string of every pair equal the oracle's traceback
    (same tie rules as the host traceback in libswsharp)."""
    rng = np.random.default_rng(29)
    qs = rand_seqs(rng, 6, 40, 300) + [rand_seqs(rng, 1, 1, 1)[0], rand_seqs(rng, 1, 33, 33)[0]]
    ts = rand_seqs(rng, 150, 1, 350)
    for k in range(60):                                   # homologs with indels and low-complexity runs (ties)
        src = qs[k % 6]
        cut = int(rng.integers(5, len(src) - 5))
        gap = int(rng.integers(1, 12))
        ts[k] = np.concatenate([src[:cut], rand_seqs(rng, 1, gap, gap)[0], src[cut + (k % 4):]]) if k % 2 else np.concatenate([src[:cut], src[cut + gap:]])
    ts[60] = np.full(120, 0, dtype=np.uint8); qs.append(np.full(90, 0, dtype=np.uint8))          # poly-A vs poly-A: massive ties
    ts[61] = np.tile(np.array([0, 2], dtype=np.uint8), 70); qs.append(np.tile(np.array([2, 0], dtype=np.uint8), 50))
    MOVE = {0: 1, 1: 2, 2: 3}
    for tile_seqs in (0, 64):
        db = s4g.Database(ts, tile_seqs=tile_seqs)
        q = s4g.QuerySet(qs)
        scores = s4g.scan_host(db, q, scorer)
        pq, pt = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
        pq, pt = pq.ravel(), pt.ravel()
        perm = rng.permutation(len(pq))                    # arbitrary pair order, > 64 survivors per query -> several items
        pq, pt = pq[perm], pt[perm]
        want = [oracle_mod.sw_align(qs[a], ts[b], blosum62, 10, 1) for a, b in zip(pq, pt)]
        # both device routes: one wave per pair (sw_pair_trace, the default for these query lengths) and the
        # 64-pairs-per-wave strip kernels (sw_trace_fill + sw_trace_walk)
        for route in (None, "wave", "scan"):
            res = s4g.traceback(db, q, scorer, pq, pt, scores[pq, pt], route=route)
            n_pos = 0
            for a, b, r, o in zip(pq, pt, res, want):
                assert o["score"] == r["score"]
                if o["score"] == 0:
                    assert len(r["path"]) == 0 and r["q_start"] == -1
                    continue
                n_pos += 1
                assert (r["q_start"], r["q_end"], r["t_start"], r["t_end"]) == (o["q_start"], o["q_end"], o["t_start"], o["t_end"]), (route, a, b)
                assert r["path"].tolist() == [MOVE[m] for m in o["path"]], (route, a, b)
            assert n_pos > 1000
    with pytest.raises(Exception):                         # a wrong score is detected, not silently traced
        s4g.traceback(db, q, scorer, [0], [0], [int(scores[0, 0]) + 1])


def test_traceback_in_two_halves_equals_the_synchronous_one(s4g, oracle_mod, blosum62, scorer):
    """s4g_traceback_submit / s4g_traceback_finish: several jobs in flight at once, finished out of order, the database handle
    destroyed between submit and finish (include/s4g_hip.h allows it), a score scan queued behind the jobs -- every job's
    coordinates and paths equal s4g_traceback_host's, which test_device_traceback_equals_oracle pins to the oracle."""
    rng = np.random.default_rng(31)
    qs = rand_seqs(rng, 12, 30, 420)
    ts = rand_seqs(rng, 400, 1, 500)
    for k in range(120):
        src = qs[k % 12]
        cut = int(rng.integers(5, len(src) - 5))
        ts[k] = np.concatenate([src[:cut], rand_seqs(rng, 1, 1 + k % 9, 1 + k % 9)[0], src[cut + (k % 3):]])
    q = s4g.QuerySet(qs)
    db = s4g.Database(ts)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    scores = s4g.scan_host(db, q, scorer)
    assert np.array_equal(scores, exp)
    sets = []
    for j in range(5):
        n = (0, 1, 64, 700, 2500)[j]
        pq = rng.integers(0, len(qs), n); pt = rng.integers(0, len(ts), n)
        sets.append((pq, pt, scores[pq, pt]))
    want = [s4g.traceback(db, q, scorer, *s) for s in sets]
    assert sum(r["score"] > 0 for w in want for r in w) > 2000
    db2 = s4g.Database(ts)                                # its tables go away while jobs that read them are in flight
    jobs = [s4g.TraceJob(db2 if j % 2 else db, q, scorer, *s) for j, s in enumerate(sets)]
    del db2
    import gc; gc.collect()
    again = s4g.scan_host(db, q, scorer)                  # the scan's stream work and the jobs' do not disturb each other
    assert np.array_equal(again, exp)
    for j in (3, 0, 4, 1, 2):
        got = jobs[j].result()
        assert len(got) == len(want[j])
        for r, o in zip(got, want[j]):
            assert (r["score"], r["q_start"], r["q_end"], r["t_start"], r["t_end"]) == (o["score"], o["q_start"], o["q_end"], o["t_start"], o["t_end"])
            assert np.array_equal(r["path"], o["path"])
    with pytest.raises(RuntimeError):
        jobs[0].result()
    # a wrong score surfaces at submit or at finish, never as a silently wrong path
    with pytest.raises(Exception):
        s4g.TraceJob(db, q, scorer, [0], [0], [int(scores[0, 0]) + 1]).result()
    # and the library is still usable after the failed job
    ok = s4g.TraceJob(db, q, scorer, sets[2][0], sets[2][1], sets[2][2]).result()
    assert all(np.array_equal(r["path"], o["path"]) for r, o in zip(ok, want[2]))


def test_packed_database_file_roundtrip(s4g, oracle_mod, blosum62, scorer, tmp_path):
    """s4g_db_save / s4g_db_load: a database placed from its packed file scans identically (full matrix, streaming
    top-k and traceback all work off the loaded tiles)."""
    rng = np.random.default_rng(30)
    qs = rand_seqs(rng, 5, 40, 200)
    ts = rand_seqs(rng, 300, 1, 280) + [qs[0][10:150].copy(), rand_seqs(rng, 1, 12400, 12400)[0]]
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    db = s4g.Database(ts, tile_seqs=128)
    path = tmp_path / "db.s4g"
    db.save(path)
    db2 = s4g.Database.load(path)
    assert db2.n == len(ts) and db2.tiles() == db.tiles() and db2.info()["residues"] == db.info()["residues"]
    q = s4g.QuerySet(qs)
    got = s4g.scan_host(db2, q, scorer)
    assert np.array_equal(got, exp)
    a = s4g.scan_topk(db, q, scorer, 20, 1); b = s4g.scan_topk(db2, q, scorer, 20, 1)
    assert all(np.array_equal(x.cpu().numpy(), y.cpu().numpy()) for x, y in zip(a, b))
    (path.parent / "junk.s4g").write_bytes(b"not a database")
    with pytest.raises(Exception):
        s4g.Database.load(path.parent / "junk.s4g")


def test_streamed_database_scans_like_the_resident_one(s4g, oracle_mod, blosum62, scorer, tmp_path):
    """s4g_db_open_streamed: the packed file scanned tile by tile through two device slots (the host reads tile i + 1 while tile i
    runs) -- the streaming top-k of 7 tiles equals the resident database's and the oracle's, twice in a row (slot reuse), with a
    long query whose score bound leaves the int16 windows (int32 tier instead of the speculative one); the byte count of the reads
    is the file's tiles; the entry points that need resident tiles refuse it."""
    rng = np.random.default_rng(130)
    qs = rand_seqs(rng, 5, 40, 200) + [rand_seqs(rng, 1, 6500, 6500)[0]]
    ts = rand_seqs(rng, 420, 1, 280) + [qs[0][10:150].copy(), qs[5][100:6400].copy(), rand_seqs(rng, 1, 12400, 12400)[0]]
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    db = s4g.Database(ts, tile_seqs=64)
    path = tmp_path / "db.s4g"
    db.save(path)
    sdb = s4g.Database.open_streamed(path)
    assert sdb.n == len(ts) and sdb.tiles() == db.tiles() and len(sdb.tiles()) >= 7
    q = s4g.QuerySet(qs)
    want = s4g.scan_topk(db, q, scorer, 25, 1)
    for rep in range(2):
        got = s4g.scan_topk(sdb, q, scorer, 25, 1)
        assert all(np.array_equal(x.cpu().numpy(), y.cpu().numpy()) for x, y in zip(got, want)), rep
    ids, sc, cnt = (x.cpu().numpy() for x in got)
    for i in range(len(qs)):
        order = sorted(range(len(ts)), key=lambda t: (-int(exp[i, t]), t))
        keep = [t for t in order if exp[i, t] >= 1][:25]
        assert ids[i, :cnt[i]].tolist() == keep and sc[i, :cnt[i]].tolist() == [int(exp[i, t]) for t in keep]
    st = sdb.stream_stats()
    assert st["tiles"] == 2 * len(sdb.tiles()) and 0 < st["bytes_read"] <= 2 * os.path.getsize(path) and st["scan_wall_s"] > 0
    with pytest.raises(RuntimeError):
        s4g.scan_host(sdb, q, scorer)
    with pytest.raises(RuntimeError):
        s4g.score_pairs(sdb, q, scorer, [0], [0])
    with pytest.raises(RuntimeError):
        sdb.save(tmp_path / "again.s4g")
    (tmp_path / "short.s4g").write_bytes(open(path, "rb").read()[:-1000])
    with pytest.raises(RuntimeError):
        s4g.Database.open_streamed(tmp_path / "short.s4g")
    sdb.close()


def test_randomized_configurations(s4g, oracle_mod):
    """seeded fuzz over everything at once: random symmetric/asymmetric tables, gap penalties, length mixes (empties,
    singletons, long tails), tile sizes and kernel choice; scan, streaming top-k and traceback against the oracle."""
    rng = np.random.default_rng(2024)
    MOVE = {0: 1, 1: 2, 2: 3}
    for it in range(24):
        M = rng.integers(-9, 12, size=(26, 26)).astype(np.int32)
        if it % 3:
            M = np.triu(M) + np.triu(M, 1).T                 # symmetric most of the time
        np.fill_diagonal(M, rng.integers(3, 14, size=26))
        go = int(rng.integers(1, 16)); ge = int(rng.integers(0, go + 1))
        sc = s4g.Scorer(M, go, ge)
        nq, nt = int(rng.integers(1, 9)), int(rng.integers(1, 260))
        qs = rand_seqs(rng, nq, 0, int(rng.choice([5, 40, 130, 400])), 26)
        ts = rand_seqs(rng, nt, 0, int(rng.choice([3, 60, 200, 700])), 26)
        if nt > 3 and len(qs[0]) > 20:                       # a few related pairs so scores are not all tiny
            ts[1] = qs[0][3:].copy()
            ts[2] = np.concatenate([qs[0][:10], rand_seqs(rng, 1, 4, 4, 26)[0], qs[0][10:]])
        exp = oracle_mod.scan(qs, ts, M, go, ge, threads=8)
        db = s4g.Database(ts, tile_seqs=int(rng.choice([0, 64, 128])))
        q = s4g.QuerySet(qs)
        mode = [dict(), dict(no_f16=True), dict(force_i32=True), dict(f16_arith=True), dict(no_drift=True), dict()][it % 6]
        got = s4g.scan_host(db, q, sc, **mode)
        assert np.array_equal(got, exp), (it, go, ge, mode)
        k = int(rng.integers(1, 50))
        ids, tsc, cnt = s4g.scan_topk(db, q, sc, k, 1, **mode)
        for i in range(nq):
            order = sorted((j for j in range(nt) if exp[i, j] >= 1), key=lambda j: (-exp[i, j], j))[:k]
            assert ids[i, :len(order)].cpu().tolist() == order and int(cnt[i]) == len(order)
        pq = rng.integers(0, nq, size=40); pt = rng.integers(0, nt, size=40)
        for a, b, r in zip(pq, pt, s4g.traceback(db, q, sc, pq, pt, exp[pq, pt])):
            o = oracle_mod.sw_align(qs[a], ts[b], M, go, ge)
            assert r["path"].tolist() == [MOVE[m] for m in o["path"]], (it, a, b)
            if o["score"]:
                assert (r["q_start"], r["q_end"], r["t_start"], r["t_end"]) == (o["q_start"], o["q_end"], o["t_start"], o["t_end"])


def test_concurrent_host_threads_and_streams(s4g, oracle_mod, blosum62, scorer):
    """the per-device scratch (workspace, counters, query lists) is shared by every database on the device: scans issued
    from several host threads, on different databases and different streams, must serialise correctly."""
    import threading
    import torch
    rng = np.random.default_rng(36)
    qs = rand_seqs(rng, 6, 40, 220)
    sets = [rand_seqs(rng, 150 + 37 * i, 5, 260) for i in range(3)]
    exps = [oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8) for ts in sets]
    dbs = [s4g.Database(ts) for ts in sets]
    q = s4g.QuerySet(qs)
    errors = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            for rep in range(6):
                with torch.cuda.stream(st):
                    out = torch.empty((len(qs), dbs[i].n), dtype=torch.int32, device="cuda:0")
                    s4g.scan(dbs[i], q, scorer, out, force_i32=bool(rep & 1))
                    ids, sc, cnt = s4g.scan_topk(dbs[i], q, scorer, 7, 1)
                st.synchronize()
                if not np.array_equal(out.cpu().numpy(), exps[i]):
                    errors.append((i, rep, "scan"))
                want = [sorted(range(dbs[i].n), key=lambda j: (-exps[i][r, j], j))[:7] for r in range(len(qs))]
                if ids.cpu().tolist() != want:
                    errors.append((i, rep, "topk"))
        except Exception as e:          # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_fresh_query_set_scanned_at_once_on_a_non_blocking_stream(s4g, oracle_mod, blosum62, scorer):
    """s4g_queries_create uploads a small set asynchronously; a scan queued right behind it on a stream of the caller's (every
    torch.cuda.Stream() is hipStreamNonBlocking: the null stream orders nothing against it) must see the NEW set -- the create call now
    waits for its own copy on its own stream.  Many short sets in a row through the recycled query blocks, each scanned at once."""
    import torch
    rng = np.random.default_rng(361)
    ts = rand_seqs(rng, 300, 5, 200)
    db = s4g.Database(ts)
    st = torch.cuda.Stream()
    busy = torch.empty((64 << 20,), dtype=torch.int32, device="cuda:0")
    for it in range(60):
        qs = rand_seqs(rng, 1 + it % 3, 20, 180)
        exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
        busy.add_(1)                                     # keeps the null stream occupied while the set is made and scanned
        q = s4g.QuerySet(qs)
        with torch.cuda.stream(st):
            out = torch.empty((len(qs), db.n), dtype=torch.int32, device="cuda:0")
            s4g.scan(db, q, scorer, out, no_pairs=bool(it & 1))
        st.synchronize()
        assert np.array_equal(out.cpu().numpy(), exp), it
        del q


def test_reduced_workspace_budget_runs_on_a_smaller_grid(s4g, oracle_mod, blosum62, scorer, monkeypatch):
    """the strip-boundary workspace budget of a call is cut to what the device has free when a tile wants more than 8 GB;
    S4G_WORKSPACE_BUDGET forces the cut: the same scores on a grid of a few blocks (long targets: 12 000 residues = 6 MB per wave slot),
    full matrix and streaming top-k, and the launch's block count (s4g_scan_clock) shrinks with the budget."""
    from sift4g_amd import device as s4gd
    rng = np.random.default_rng(362)
    qs = rand_seqs(rng, 24, 300, 400)
    ts = rand_seqs(rng, 3000, 30, 400) + rand_seqs(rng, 70, 9000, 12000)
    ts[3] = qs[2].copy()
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=os.cpu_count() or 8)
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    assert np.array_equal(s4g.scan_host(db, q, scorer, no_pairs=True), exp)
    blocks_full = s4gd.scan_clock(0)["blocks"]                             # blocks of the longest launch that found work (s4g_scan_clock)
    monkeypatch.setenv("S4G_WORKSPACE_BUDGET", str(200 << 20))             # 24.6 MB per block of four waves at 12 000 residues: 8 blocks
    assert np.array_equal(s4g.scan_host(db, q, scorer, no_pairs=True), exp)
    blocks_small = s4gd.scan_clock(0)["blocks"]
    assert 1 <= blocks_small <= 8 < blocks_full, (blocks_small, blocks_full)
    for mode in (dict(no_pairs=True), dict(no_pairs=True, force_i32=True), dict(no_pairs=True, diag_frame=True)):
        assert np.array_equal(s4g.scan_host(db, q, scorer, **mode), exp), mode
    ids, sc, cnt = [x.cpu().numpy() for x in s4g.scan_topk(db, q, scorer, 9, 1)]
    for i in range(len(qs)):
        order = sorted(range(len(ts)), key=lambda j: (-exp[i, j], j))[:9]
        assert ids[i].tolist() == order and sc[i].tolist() == [int(exp[i, j]) for j in order]


def test_argument_errors_are_reported(s4g, scorer):
    """bad arguments come back as error codes with text (never a crash, never a silent fallback)."""
    import torch
    from sift4g_amd._lib import S4GError
    rng = np.random.default_rng(37)
    db = s4g.Database(rand_seqs(rng, 70, 5, 60))
    q = s4g.QuerySet(rand_seqs(rng, 3, 20, 50))
    out = torch.empty((3, db.n), dtype=torch.int32, device="cuda:0")
    with pytest.raises(S4GError, match="out of bounds"):
        s4g.scan(db, q, scorer, torch.empty((5, db.n), dtype=torch.int32, device="cuda:0"), q_begin=2, q_count=5)
    with pytest.raises(S4GError, match="k must be"):
        s4g.scan_topk(db, q, scorer, 5000)
    with pytest.raises(S4GError, match="k must be"):
        s4g.topk(db, out, 0)
    with pytest.raises(S4GError, match="out of range"):
        s4g.Database([np.array([3, 26, 1], dtype=np.uint8)])
    with pytest.raises(S4GError, match="out of range"):
        s4g.QuerySet([np.array([31], dtype=np.uint8)])
    with pytest.raises(S4GError, match="out of range"):
        s4g.traceback(db, q, scorer, [0], [db.n], [5])
    with pytest.raises(S4GError, match="device"):
        s4g.Database(rand_seqs(rng, 2, 5, 9), device=63)
    with pytest.raises(ValueError):
        s4g.scan(db, q, scorer, out.to(torch.int64))


# ---------------------------------------------------------------------------------------------- pair lists (sw_pair.hip.h)
def _pairs_check(s4g, oracle_mod, M, qs, ts, pq, pt, sc, go=10, ge=1, routes=("wave", None, "scan")):
    qc, qo = oracle_mod._flat(qs)
    tc, to = oracle_mod._flat(ts)
    want = oracle_mod.pairs_flat(qc, qo, tc, to, pq, pt, M, go, ge, threads=8)
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    for route in routes:
        got = s4g.score_pairs(db, q, sc, pq, pt, route=route)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, (f"route {route}: {len(bad)} mismatches, first pairs {bad[:5].tolist()} "
                               f"(lq {len(qs[pq[bad[0]]])}, lt {len(ts[pt[bad[0]]])}) got {got[bad[0]]} want {want[bad[0]]}")


def test_pairs_every_row_block_class(s4g, oracle_mod, blosum62, scorer):
    """one wave per pair: query lengths on both sides of every rows-per-lane class (64 x {2,4,6,8,12,16} rows) and
    target lengths around the 4-column dword and the 64-lane skew."""
    rng = np.random.default_rng(31)
    qlens = [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 385, 511, 512, 513, 767, 768, 769, 1023, 1024]
    tlens = [1, 2, 3, 4, 5, 62, 63, 64, 65, 66, 127, 129, 300, 777]
    qs = [rand_seqs(rng, 1, l, l)[0] for l in qlens]
    ts = [rand_seqs(rng, 1, l, l)[0] for l in tlens]
    pq, pt = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
    _pairs_check(s4g, oracle_mod, blosum62, qs, ts, pq.ravel(), pt.ravel(), scorer)


def test_pairs_random_lists_all_codes(s4g, oracle_mod, blosum62, scorer):
    """scattered candidate lists (with repeats) over a database with all 26 codes, homologs planted."""
    rng = np.random.default_rng(32)
    qs = rand_seqs(rng, 12, 30, 700, 26)
    ts = rand_seqs(rng, 900, 1, 900, 26)
    for i in range(0, 900, 37):                              # high-scoring pairs: a mutated copy of some query
        src = qs[i % len(qs)].copy()
        src[rng.integers(0, len(src), size=max(1, len(src) // 10))] = rng.integers(0, 20, size=max(1, len(src) // 10))
        ts[i] = np.delete(src, rng.integers(0, len(src), size=3))
    n = 4000
    pq = rng.integers(0, len(qs), size=n)
    pt = rng.integers(0, len(ts), size=n)
    _pairs_check(s4g, oracle_mod, blosum62, qs, ts, pq, pt, scorer)


def test_pairs_target_windows_and_long_queries(s4g, oracle_mod, blosum62, scorer):
    """targets longer than the LDS window of the pair kernels (kPairWindow columns: several windows, lengths on both sides
    of the boundaries, odd step counts); queries beyond 1024 residues run band by band (sw_pair_scan_band)."""
    rng = np.random.default_rng(33)
    qs = [rand_seqs(rng, 1, l, l)[0] for l in (97, 350, 640, 1024, 1025, 2300)]
    ts = [rand_seqs(rng, 1, l, l)[0] for l in (4095, 4096, 4097, 8191, 8200, 12289, 13001, 40, 1, 1985, 2047, 2048, 2049, 2112)]
    ts[4][100:450] = qs[1]                                   # a real hit deep inside the second window
    ts[6][8100:8740] = qs[2]
    pq, pt = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
    _pairs_check(s4g, oracle_mod, blosum62, qs, ts, pq.ravel(), pt.ravel(), scorer, routes=(None, "scan", "wave"))


def test_pairs_empty_sequences_and_other_scorers(s4g, oracle_mod, blosum62, scorer):
    rng = np.random.default_rng(34)
    e = np.zeros(0, dtype=np.uint8)
    qs = [e, rand_seqs(rng, 1, 90, 90)[0], rand_seqs(rng, 1, 300, 300)[0]]
    ts = [rand_seqs(rng, 1, 120, 120)[0], e, rand_seqs(rng, 1, 1000, 1000)[0]]
    pq, pt = np.meshgrid(np.arange(3), np.arange(3), indexing="ij")
    _pairs_check(s4g, oracle_mod, blosum62, qs, ts, pq.ravel(), pt.ravel(), scorer)
    assert len(s4g.score_pairs(s4g.Database(ts), s4g.QuerySet(qs), scorer, [], [])) == 0
    for go, ge in ((1, 1), (12, 3), (40, 40)):
        _pairs_check(s4g, oracle_mod, blosum62, qs, ts, pq.ravel(), pt.ravel(), s4g.Scorer("BLOSUM_62", go, ge), go, ge)
    M = rng.integers(-30, 60, size=(26, 26)).astype(np.int32)     # large, asymmetric scores: int32 range is exercised
    M[:, 3] = 55
    _pairs_check(s4g, oracle_mod, M, qs, ts, pq.ravel(), pt.ravel(), s4g.Scorer(M, 7, 2), 7, 2)
    with pytest.raises(RuntimeError):
        s4g.score_pairs(s4g.Database(ts), s4g.QuerySet(qs), scorer, [3], [0])


def test_packed_pair_kernel_takes_what_its_window_admits(s4g, oracle_mod, blosum62):
    """sw_pair_scan_pk (two rows per register, the upper halves one column behind; sw_pair_pk.hip.h) serves the pairs of queries of up to
    384 residues whose drifting frame fits the int16 window; the rest of the same call -- longer queries, targets beyond the window's
    reach -- stays on the int32 kernel.  Query lengths on both sides of its three classes (128 / 256 / 384 rows), targets of one
    residue to several LDS windows, exact copies (the highest scores a pair can have), a scorer whose extension penalty of 4 cuts
    the reach to a few thousand columns (so that one class is split between the two kernels), and one whose penalties admit nothing:
    every score against the oracle, and the routing statistics say who ran where."""
    from sift4g_amd import device as s4gd
    rng = np.random.default_rng(91)
    qlens = [1, 2, 7, 64, 127, 128, 129, 200, 255, 256, 257, 300, 350, 383, 384, 385, 400, 512]
    tlens = [1, 2, 3, 5, 63, 64, 65, 127, 128, 129, 130, 340, 1000, 2047, 2048, 2049, 2175, 2176, 2177, 4500, 7000, 9000, 20000, 29000, 33000]
    qs = [rand_seqs(rng, 1, l, l)[0] for l in qlens]
    ts = [rand_seqs(rng, 1, l, l)[0] for l in tlens]
    ts += [q.copy() for q in qs]                                            # self-scores: the top of the window
    ts.append(np.concatenate([rand_seqs(rng, 1, 2040, 2040)[0], qs[12], rand_seqs(rng, 1, 300, 300)[0]]))      # a hit across a window boundary
    pq, pt = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
    pq, pt = pq.ravel(), pt.ravel()
    n_pk_queries = sum(1 for l in qlens if l <= 384)
    for (go, ge), expect in (((10, 1), "most"), ((12, 4), "split"), ((40000, 1), "none")):
        sc = s4g.Scorer("BLOSUM_62", go, ge)
        _pairs_check(s4g, oracle_mod, blosum62, qs, ts, pq, pt, sc, go, ge, routes=("wave",))
        st = s4gd.scan_stats(0)
        assert st["pair_items"] == len(pq)
        if expect == "most":          # everything of the short queries but the 33 000-residue target (frame beyond 0x7BFF)
            assert n_pk_queries * (len(ts) - 2) <= st["pair_pk"] <= n_pk_queries * (len(ts) - 1), st
        elif expect == "split":       # reach of a few thousand columns: the long targets of every class stay behind
            assert n_pk_queries * 20 <= st["pair_pk"] <= n_pk_queries * (len(ts) - 3), st
        else:
            assert st["pair_pk"] == 0, st


def test_pair_traceback_row_block_classes_windows_and_mixed_routes(s4g, oracle_mod, blosum62, scorer):
    """sw_pair_trace over every rows-per-lane class, targets beyond one LDS window, gap-rich alignments; queries beyond
    1024 residues in the same call run band by band (sw_pair_trace_band)."""
    rng = np.random.default_rng(35)
    qlens = [1, 2, 64, 65, 128, 129, 200, 256, 257, 384, 385, 512, 513, 700, 768, 769, 1024, 1025, 1500]
    qs = [rand_seqs(rng, 1, l, l)[0] for l in qlens]
    ts = []
    for k, src in enumerate(qs):                              # two gapped homologs per query + noise
        if len(src) > 20:
            cut = len(src) // 2
            ts.append(np.concatenate([rand_seqs(rng, 1, 30, 30)[0], src[:cut], rand_seqs(rng, 1, 7, 7)[0], src[cut:]]))
            ts.append(np.concatenate([src[:cut], src[cut + 5:], rand_seqs(rng, 1, 11, 11)[0]]))
        else:
            ts.append(src.copy())
    ts += rand_seqs(rng, 10, 1, 500)
    long_t = rand_seqs(rng, 1, 9000, 9000)[0]
    long_t[4000:4700] = qs[13]                                 # hit straddling the first window boundary (4096)
    long_t[8100:8484] = qs[9]
    ts.append(long_t)
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    scores = s4g.scan_host(db, q, scorer)
    pq, pt = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
    pq, pt = pq.ravel(), pt.ravel()
    MOVE = {0: 1, 1: 2, 2: 3}
    res = s4g.traceback(db, q, scorer, pq, pt, scores[pq, pt])
    n_pos = 0
    for a, b, r in zip(pq, pt, res):
        o = oracle_mod.sw_align(qs[a], ts[b], blosum62, 10, 1)
        assert o["score"] == r["score"], (a, b)
        if o["score"] == 0:
            assert len(r["path"]) == 0 and r["q_start"] == -1
            continue
        n_pos += 1
        assert (r["q_start"], r["q_end"], r["t_start"], r["t_end"]) == (o["q_start"], o["q_end"], o["t_start"], o["t_end"]), (a, b)
        assert r["path"].tolist() == [MOVE[m] for m in o["path"]], (a, b)
    assert n_pos > 500
    for go, ge in ((1, 1), (0, 0), (5, 5)):                  # cheap gaps: long gap runs and many ties
        sc = s4g.Scorer("BLOSUM_62", go, ge)
        sub = rng.choice(len(pq), size=150, replace=False)
        got_scores = s4g.score_pairs(db, q, sc, pq[sub], pt[sub])
        for a, b, r in zip(pq[sub], pt[sub], s4g.traceback(db, q, sc, pq[sub], pt[sub], got_scores)):
            o = oracle_mod.sw_align(qs[a], ts[b], blosum62, go, ge)
            assert o["score"] == r["score"], (go, ge, a, b)
            assert r["path"].tolist() == [MOVE[m] for m in o["path"]], (go, ge, a, b)


def test_pair_kernels_fuzz(s4g, oracle_mod, blosum62):
    """random pair lists over the whole (query length <= 1024) x (target length <= 6000) rectangle and three scorers:
    scores of sw_pair_scan and alignments of sw_pair_trace against the oracle."""
    rng = np.random.default_rng(36)
    MOVE = {0: 1, 1: 2, 2: 3}
    for go, ge in ((10, 1), (3, 3), (25, 2)):
        sc = s4g.Scorer("BLOSUM_62", go, ge)
        qlens = rng.integers(1, 1025, size=40)
        tlens = np.concatenate([rng.integers(1, 700, size=300), rng.integers(700, 6001, size=40)])
        qs = [rand_seqs(rng, 1, int(l), int(l), 26)[0] for l in qlens]
        ts = [rand_seqs(rng, 1, int(l), int(l), 26)[0] for l in tlens]
        for k in range(0, len(ts), 5):                       # related pairs: a window of some query, mutated, inside noise
            src = qs[k % len(qs)]
            a = int(rng.integers(0, max(1, len(src) // 2)))
            piece = src[a:a + max(1, len(ts[k]) // 2)].copy()
            if len(piece) > 8:
                piece[rng.integers(0, len(piece), size=len(piece) // 8)] = rng.integers(0, 20, size=len(piece) // 8)
            at = int(rng.integers(0, max(1, len(ts[k]) - len(piece) + 1)))
            ts[k][at:at + len(piece)] = piece[:len(ts[k]) - at]
        db, q = s4g.Database(ts, tile_seqs=int(rng.choice([0, 64]))), s4g.QuerySet(qs)
        n = 3000
        pq = rng.integers(0, len(qs), size=n)
        pt = rng.integers(0, len(ts), size=n)
        pq[:len(ts)] = np.arange(len(ts)) % len(qs); pt[:len(ts)] = np.arange(len(ts))      # every target at least once
        qc, qo = oracle_mod._flat(qs)
        tc, to = oracle_mod._flat(ts)
        want = oracle_mod.pairs_flat(qc, qo, tc, to, pq, pt, blosum62, go, ge, threads=8)
        got = s4g.score_pairs(db, q, sc, pq, pt, route="wave")
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, (go, ge, bad[:5].tolist(), got[bad[:5]].tolist(), want[bad[:5]].tolist())
        sub = np.concatenate([np.arange(len(ts))[::3], rng.choice(n, size=200, replace=False)])
        res = s4g.traceback(db, q, sc, pq[sub], pt[sub], want[sub], route="wave")
        for a, b, r in zip(pq[sub], pt[sub], res):
            o = oracle_mod.sw_align(qs[a], ts[b], blosum62, go, ge)
            assert o["score"] == r["score"]
            assert r["path"].tolist() == [MOVE[m] for m in o["path"]], (go, ge, int(a), int(b))
            if o["score"]:
                assert (r["q_start"], r["q_end"], r["t_start"], r["t_end"]) == (o["q_start"], o["q_end"], o["t_start"], o["t_end"])


def test_device_side_packing(s4g, oracle_mod, blosum62, scorer):
    """databases of 8 MB of residues and more are transposed on the device (sw_pack_tile): same scores as the oracle,
    single- and multi-tile, empty and 1-residue sequences, lengths around the dword; an invalid code is still refused."""
    rng = np.random.default_rng(37)
    lens = np.concatenate([rng.integers(0, 9, size=500), rng.integers(9, 700, size=33000), rng.integers(700, 2500, size=300)])
    rng.shuffle(lens)
    ts = [rand_seqs(rng, 1, int(l), int(l), 26)[0] if l else np.zeros(0, np.uint8) for l in lens]
    assert sum(len(t) for t in ts) > (8 << 20)
    qs = rand_seqs(rng, 3, 60, 420, 26)
    ts[123] = qs[0][5:300].copy()
    tc, to = oracle_mod._flat(ts)
    qc, qo = oracle_mod._flat(qs)
    exp = oracle_mod.scan_simd_flat(qc, qo, tc, to, blosum62, 10, 1, threads=8)
    if exp is None:
        exp = oracle_mod.scan_flat(qc, qo, tc, to, blosum62, 10, 1, threads=8)
    q = s4g.QuerySet(qs)
    for tile_seqs in (0, 4096):
        db = s4g.Database(ts, tile_seqs=tile_seqs)
        for mode in (dict(no_pairs=True), dict(force_i32=True), dict()):
            got = s4g.scan_host(db, q, scorer, **mode)
            bad = np.argwhere(got != exp)
            assert bad.size == 0, (tile_seqs, mode, bad[:5].tolist())
        pt = rng.integers(0, len(ts), size=300)
        r = s4g.traceback(db, q, scorer, np.zeros(300, np.uint32), pt, exp[0, pt])
        for b, x in zip(pt[:40], r[:40]):
            o = oracle_mod.sw_align(qs[0], ts[b], blosum62, 10, 1)
            assert o["score"] == x["score"] and len(o["path"]) == len(x["path"])
    ts[77] = ts[77].copy() if len(ts[77]) else np.zeros(5, np.uint8)
    ts[77][0] = 26
    with pytest.raises(RuntimeError):
        s4g.Database(ts)


def test_database_from_device_resident_codes(s4g, oracle_mod, blosum62, scorer):
    """s4g_db_create_from_device: codes generated / held on the GPU are packed there (only the offsets visit the host);
    small and multi-tile databases, same scores and alignments as the host-created database."""
    import torch
    rng = np.random.default_rng(38)
    qs = rand_seqs(rng, 5, 30, 300, 26)
    for nt, tile_seqs in ((700, 0), (700, 128), (3, 0)):
        ts = rand_seqs(rng, nt, 0, 400, 26)
        ts[1] = qs[0][3:].copy()
        tc, to = oracle_mod._flat(ts)
        d_codes = torch.from_numpy(tc.copy()).to("cuda:0")
        d_off = torch.from_numpy(to.astype(np.int64)).to("cuda:0")
        db = s4g.Database.from_device(d_codes, d_off, tile_seqs=tile_seqs)
        del d_codes, d_off                                     # the database keeps nothing of the caller's buffers
        torch.cuda.empty_cache()
        exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
        q = s4g.QuerySet(qs)
        for mode in (dict(no_pairs=True), dict()):
            assert np.array_equal(s4g.scan_host(db, q, scorer, **mode), exp), (nt, tile_seqs, mode)
        r = s4g.traceback(db, q, scorer, [0], [1], [int(exp[0, 1])])[0]
        o = oracle_mod.sw_align(qs[0], ts[1], blosum62, 10, 1)
        assert (r["q_start"], r["q_end"], r["t_start"], r["t_end"]) == (o["q_start"], o["q_end"], o["t_start"], o["t_end"])
    bad = torch.full((10,), 27, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(RuntimeError):
        s4g.Database.from_device(bad, torch.tensor([0, 10], dtype=torch.int64, device="cuda:0"))
    with pytest.raises(ValueError):
        s4g.Database.from_device(bad.cpu(), torch.tensor([0, 10], dtype=torch.int64))


def test_database_from_separate_arrays(s4g, oracle_mod, blosum62, scorer):
    """s4g_db_create_scattered (what chainDatabaseCreate passes: one code array per chain): same scores, pair scores and
    alignments as the contiguous form, with empty sequences, for a small database (host-packed out of the arrays) and one
    beyond the device-packing threshold; bad codes and NULL arguments are refused."""
    MOVE = {0: 1, 1: 2, 2: 3}
    rng = np.random.default_rng(52)
    qs = rand_seqs(rng, 3, 40, 330, 26)
    q = s4g.QuerySet(qs)
    for nt, lo, hi in ((900, 0, 500), (1, 7, 7), (0, 0, 0), (9000, 900, 1100)):
        ts = rand_seqs(rng, nt, lo, hi, 26) if nt else []
        if nt > 2:
            ts[2] = qs[1][5:].copy()
        db = s4g.Database.from_arrays(ts)
        assert db.n == nt and db.info()["residues"] == sum(len(t) for t in ts)
        if nt == 0:
            continue
        ref = s4g.Database(ts)
        exp = s4g.scan_host(ref, q, scorer, no_pairs=True)
        if nt <= 900:
            assert np.array_equal(exp, oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8))
        for mode in (dict(no_pairs=True), dict()):
            assert np.array_equal(s4g.scan_host(db, q, scorer, **mode), exp), (nt, mode)
        pt = rng.integers(0, nt, size=200).astype(np.uint32)
        pq = rng.integers(0, 3, size=200).astype(np.uint32)
        assert np.array_equal(s4g.score_pairs(db, q, scorer, pq, pt), exp[pq, pt])
        if nt > 2:
            r = s4g.traceback(db, q, scorer, [1], [2], [int(exp[1, 2])])[0]
            o = oracle_mod.sw_align(qs[1], ts[2], blosum62, 10, 1)
            assert r["score"] == o["score"] and r["path"].tolist() == [MOVE[m] for m in o["path"]]
    bad = rand_seqs(rng, 10, 5, 50, 26)
    bad[4][3] = 26
    with pytest.raises(RuntimeError):
        s4g.Database.from_arrays(bad)


def test_traceback_tie_switch_on_the_device(s4g, oracle_mod, blosum62, monkeypatch):
    """S4G_TRACEBACK_TIES (divergence manifest): pairs whose optimal alignment differs between "UP before LEFT"
    (default) and "LEFT before UP" -- both device traceback implementations follow the switch exactly as the oracle does."""
    MOVE = {0: 1, 1: 2, 2: 3}
    rng = np.random.default_rng(101)
    pairs = []
    for _ in range(20000):
        q = rand_seqs(rng, 1, 6, 40)[0]
        cut = int(rng.integers(2, len(q) - 2))
        t = np.concatenate([q[:cut], rand_seqs(rng, 1, 0, 3)[0], q[cut + int(rng.integers(0, 3)):]])
        if len(t) == 0:
            continue
        monkeypatch.setenv("S4G_TRACEBACK_TIES", "up")
        a = oracle_mod.sw_align(q, t, blosum62, 3, 3)
        monkeypatch.setenv("S4G_TRACEBACK_TIES", "left")
        b = oracle_mod.sw_align(q, t, blosum62, 3, 3)
        if a["score"] > 0 and a["path"].tolist() != b["path"].tolist():
            pairs.append((q, t))
            if len(pairs) == 12:
                break
    assert len(pairs) >= 6
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    sc = s4g.Scorer("BLOSUM_62", 3, 3)
    idx = np.arange(len(pairs), dtype=np.uint32)
    scores = s4g.score_pairs(db, q, sc, idx, idx)
    differ = 0
    paths = {}
    for mode in ("up", "left"):
        monkeypatch.setenv("S4G_TRACEBACK_TIES", mode)
        for route in ("wave", "scan"):
            res = s4g.traceback(db, q, sc, idx, idx, scores, route=route)
            for k, r in enumerate(res):
                o = oracle_mod.sw_align(qs[k], ts[k], blosum62, 3, 3)
                assert r["score"] == o["score"] and r["path"].tolist() == [MOVE[m] for m in o["path"]], (mode, route, k)
                paths[(mode, route, k)] = r["path"].tolist()
    for k in range(len(pairs)):
        differ += paths[("up", "wave", k)] != paths[("left", "wave", k)]
    assert differ == len(pairs)
    monkeypatch.delenv("S4G_TRACEBACK_TIES")


def test_candidate_databases_out_of_cached_chains(s4g, oracle_mod, blosum62, scorer):
    """s4g_db_create_cached (what libswsharp's per-query chainDatabaseCreate turns into): overlapping candidate lists out of
    one pool of chains -- every chain is sent once (the cache words the caller keeps with a chain say where it lives), scores
    and alignments equal the oracle's, long / empty / single-residue chains included; the scan entry points refuse such a
    database, a bad residue code is refused and leaves the words untouched."""
    from sift4g_amd import device as s4gd
    MOVE = {0: 1, 1: 2, 2: 3}
    rng = np.random.default_rng(77)
    pool = rand_seqs(rng, 600, 1, 700, 26) + [rand_seqs(rng, 1, 5000, 5000, 26)[0], np.zeros(0, np.uint8), rand_seqs(rng, 1, 1, 1, 26)[0]]
    qs = rand_seqs(rng, 3, 60, 400, 26) + [rand_seqs(rng, 1, 1300, 1300, 26)[0]]
    pool[7] = np.concatenate([qs[1][:150], qs[1][160:]])
    pool[600] = np.concatenate([pool[600][:2000], qs[3], pool[600][2000:3700]])        # a banded query inside the long chain
    q = s4g.QuerySet(qs)
    tags, addrs = np.zeros(len(pool), np.uint64), np.zeros(len(pool), np.uint64)
    before = s4gd.chain_cache_info(0)
    seen = set()
    for it in range(6):
        ids = np.sort(rng.choice(len(pool), size=250, replace=False))
        if it == 0:
            ids = np.union1d(ids, [7, 600, 601, 602])
        t, a = tags[ids].copy(), addrs[ids].copy()
        db = s4g.Database.from_cached([pool[i] for i in ids], t, a)
        assert np.all(t != 0) and np.all(a[[len(pool[i]) > 0 for i in ids]] != 0)
        assert np.array_equal(t[tags[ids] != 0], tags[ids][tags[ids] != 0]) and np.array_equal(a[tags[ids] != 0], addrs[ids][tags[ids] != 0])
        tags[ids], addrs[ids] = t, a
        info = s4gd.chain_cache_info(0)
        new = len(set(ids.tolist()) - seen)
        seen |= set(ids.tolist())
        assert info["misses"] - before["misses"] == len(seen) and info["alive"] == before["alive"] + 1, (info, before, new)
        pq = np.repeat(np.arange(len(qs), dtype=np.uint32), len(ids))
        pt = np.tile(np.arange(len(ids), dtype=np.uint32), len(qs))
        got = s4g.score_pairs(db, q, scorer, pq, pt).reshape(len(qs), len(ids))
        exp = oracle_mod.scan(qs, [pool[i] for i in ids], blosum62, 10, 1, threads=8)
        assert np.array_equal(got, exp), it
        assert np.array_equal(s4g.score_pairs(db, q, scorer, pq, pt, route="lists").reshape(len(qs), len(ids)), exp)     # (forced back to waves)
        top = np.argsort(-exp, axis=1)[:, :6]
        tq = np.repeat(np.arange(len(qs)), 6); tt = top.ravel()
        for r, a_, b_ in zip(s4g.traceback(db, q, scorer, tq, tt, exp[tq, tt], route="scan"), tq, tt):
            o = oracle_mod.sw_align(qs[a_], pool[ids[b_]], blosum62, 10, 1)
            assert r["score"] == o["score"] and r["path"].tolist() == [MOVE[m] for m in o["path"]], (it, a_, b_)
        with pytest.raises(RuntimeError):
            s4g.scan_host(db, q, scorer)
        db.close()
    assert s4gd.chain_cache_info(0)["alive"] == before["alive"]
    bad = [pool[3].copy(), pool[4].copy()]
    bad[1][0] = 26
    t, a = np.zeros(2, np.uint64), np.zeros(2, np.uint64)
    with pytest.raises(RuntimeError):
        s4g.Database.from_cached(bad, t, a)
    assert not t.any() and not a.any()


def test_speculative_drift_tier_for_long_queries(s4g, oracle_mod, blosum62):
    """Queries whose score BOUND is beyond the drifting-frame window (here 2048 .. 5000 residues) still run in sw_scan_pkd,
    speculatively: results that come near the window's end are marked by the kernel and recomputed in int32 by
    sw_pair_recheck before anything reads them.  Targets: unrelated ones (far below the mark), whole copies and close
    homologs of the queries (self-scores of 11k .. 27k: above and below the mark, in both halves of a packed pair), with the
    default gaps, with extend = 2 (a frame that drifts twice as fast: lower mark) and with extend = 5 (no drift frame fits:
    speculative maximum3 tier).  Scan and streaming top-k, against
    the oracle and against the tiers-by-bound routing (no_spec)."""
    rng = np.random.default_rng(61)
    qlens = [2500, 3300, 4100, 5000, 4999, 2048, 900, 6000]
    qs = [rand_seqs(rng, 1, l, l)[0] for l in qlens]
    ts = rand_seqs(rng, 150, 30, 900)
    ts[60] = qs[7].copy()                          # self-score ~35 000: beyond int16 altogether, marked in every packed tier
    def mutate(src, frac, indel):
        x = src.copy()
        idx = rng.integers(0, len(x), size=int(len(x) * frac))
        x[idx] = rand_seqs(rng, 1, len(idx), len(idx))[0]
        if indel:
            cut = int(rng.integers(100, len(x) - 100))
            x = np.concatenate([x[:cut], rand_seqs(rng, 1, indel, indel)[0], x[cut + 7:]])
        return x
    ts[3] = qs[3].copy(); ts[9] = qs[4].copy(); ts[17] = qs[0].copy(); ts[30] = qs[2].copy(); ts[31] = qs[6].copy()
    ts[40] = mutate(qs[3], 0.05, 0); ts[41] = mutate(qs[3], 0.15, 12); ts[42] = mutate(qs[1], 0.02, 5)
    ts[50] = np.concatenate([rand_seqs(rng, 1, 300, 300)[0], qs[4][1000:4800], rand_seqs(rng, 1, 200, 200)[0]])
    ts[51] = np.concatenate([qs[5], qs[5]])
    q = s4g.QuerySet(qs)
    # (12, 5): a frame that no longer fits the drift window at these lengths -> the same scheme in the fixed-frame maximum3 tier
    for go, ge in ((10, 1), (11, 2), (12, 5)):
        sc = s4g.Scorer("BLOSUM_62", go, ge)
        exp = oracle_mod.scan(qs, ts, blosum62, go, ge, threads=8)
        assert exp.max() > 32767 and np.sort(exp[3])[-3] > 12000
        for tile_seqs in (0, 64):
            db = s4g.Database(ts, tile_seqs=tile_seqs)
            # block_pipe (round 6): the block-cooperative strip pipeline takes the queries beyond its window speculatively too (its strips
            # mark what they cannot vouch for, sw_pair_recheck recomputes); with no_spec it leaves them to the tiers by bound
            for mode in (dict(no_pairs=True), dict(diag_frame=True), dict(no_spec=True), dict(no_spec=True, diag_frame=True), dict(),
                         dict(block_pipe=True), dict(block_pipe=True, no_spec=True)):
                got = s4g.scan_host(db, q, sc, **mode)
                bad = np.argwhere(got != exp)
                assert bad.size == 0, (go, ge, tile_seqs, mode, bad[:5].tolist(), got[tuple(bad[0])], exp[tuple(bad[0])])
            ids, top, cnt = s4g.scan_topk(db, q, sc, 8, no_pairs=True)
            ids, top, cnt = ids.cpu().numpy(), top.cpu().numpy(), cnt.cpu().numpy()
            for i in range(len(qs)):
                order = sorted(range(len(ts)), key=lambda t: (-int(exp[i, t]), t))[:8]
                assert ids[i].tolist() == order and top[i].tolist() == [int(exp[i, t]) for t in order], (go, ge, tile_seqs, i)


def test_speculative_launch_really_marks_results(s4g, oracle_mod, blosum62):
    """the same kind of input, looking at what the library says it did (s4g_scan_stats): the long queries must have been routed
    speculatively, the kernel must have marked results -- the self hits beyond its limit -- and the exact int32 pass must have
    recomputed exactly those; with routing by bound (no_spec) nothing is speculative and nothing is marked.  (Round 1 switched
    the recheck off through an environment hook inside the product library to see the marks; the hook is gone.)"""
    from sift4g_amd import device as s4gd
    rng = np.random.default_rng(62)
    qs = [rand_seqs(rng, 1, l, l)[0] for l in (5000, 4200, 2600, 2500)]
    ts = rand_seqs(rng, 70, 30, 600)
    ts[5] = qs[0].copy(); ts[6] = qs[1].copy(); ts[7] = qs[2].copy(); ts[8] = qs[3][:2400].copy()
    sc = s4g.Scorer("BLOSUM_62", 10, 1)
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    got = s4g.scan_host(db, q, sc, diag_frame=True)
    st = s4gd.scan_stats(0)
    assert np.array_equal(got, exp)
    # anti-diagonal frame (sw_scan_pkd): self hits of the 5000- and 4200-residue queries (29k, 24k) lie beyond the limit
    # (31 666 - the largest frame value of the pair: rows + columns); 2600 / 2500 (15k, 14k) do not
    assert st["speculative"] >= 2 and 2 <= st["rechecked"] <= 8 and st["tier_i32"] == 0, st
    got = s4g.scan_host(db, q, sc, no_pairs=True)
    st = s4gd.scan_stats(0)
    assert np.array_equal(got, exp)
    # column frame (sw_scan_pkc): its window has no row term, so only the 5000-residue query (bound 28k + 64 + 5008 columns) is
    # beyond it; the mark is taken IN the frame, score + B + column, against 0x7BFF - 12: the 29k self hit ends at column 5000
    # (34k: marked, and its partner's result at that target with it)
    assert st["speculative"] >= 1 and 1 <= st["rechecked"] <= 4 and st["tier_i32"] == 0, st
    got = s4g.scan_host(db, q, sc, no_pairs=True, no_spec=True)
    st = s4gd.scan_stats(0)
    assert np.array_equal(got, exp) and st["speculative"] == 0 and st["rechecked"] == 0, st
    # the block pipeline (round 6) takes tiles of targets up to 4096 residues: here the 5000-residue copy keeps the tile on the general
    # route -- the same matrix; test_speculative_block_pipeline_marks_and_rechecks looks at a tile it does take
    assert np.array_equal(s4g.scan_host(db, q, sc, block_pipe=True), exp)
    # extend = 5: no drift frame fits at these lengths -> the fixed-frame maximum3 tier, speculative for the long queries;
    # its window is on the score itself: only the 6000-residue self hit (~35 000) passes 0x7BFF - bias
    qs2 = qs[:2] + [rand_seqs(rng, 1, 6000, 6000)[0]]
    ts2 = ts[:20] + [qs2[2].copy()]
    exp2 = oracle_mod.scan(qs2, ts2, blosum62, 12, 5, threads=8)
    got2 = s4g.scan_host(s4g.Database(ts2), s4g.QuerySet(qs2), s4g.Scorer("BLOSUM_62", 12, 5), no_pairs=True)
    st = s4gd.scan_stats(0)
    assert np.array_equal(got2, exp2) and exp2[2, 20] > 32767
    # (the upper half of a register pair is marked together with a marked lower half: the partner query's result at that target too)
    assert st["speculative"] >= 1 and 1 <= st["rechecked"] <= 2 and st["tier_m3"] >= 1, st


def test_speculative_block_pipeline_marks_and_rechecks(s4g, oracle_mod, blosum62):
    """Round 6: the block-cooperative strip pipeline (sw_scan_lists in scan mode) runs queries whose score BOUND is beyond its window
    speculatively, like sw_scan_pkc: each strip's in-frame running maximum is its witness, a strip that cannot vouch for its values marks
    the item's result, sw_pair_recheck recomputes the marked entries in int32.  Queries of 5000 / 4200 / 2500 random residues and 4500
    tryptophans (11 per match: bound 49 500) against short targets, a 4000-residue piece of the first query (23 k: inside the window) and
    3500 tryptophans (38 500: beyond int16 altogether).  The whole call runs on the pipeline, speculatively; exactly the entries at the
    tryptophan target that share a register pair with the tryptophan query are marked and recomputed; with no_spec the pipeline leaves the
    tile to the tiers by bound; a call of ONE such query takes the same route."""
    from sift4g_amd import device as s4gd
    rng = np.random.default_rng(63)
    qs = [rand_seqs(rng, 1, l, l)[0] for l in (5000, 4200)] + [np.full(4500, 22, dtype=np.uint8), rand_seqs(rng, 1, 2500, 2500)[0]]
    ts = rand_seqs(rng, 70, 30, 600)
    ts[5] = qs[0][500:4500].copy(); ts[6] = np.full(3500, 22, dtype=np.uint8); ts[7] = qs[3][:2400].copy()
    ts[8] = np.concatenate([qs[1][:2000], np.full(600, 22, dtype=np.uint8), qs[1][2000:3000]])
    sc = s4g.Scorer("BLOSUM_62", 10, 1)
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    assert exp[2, 6] == 38500 and exp[0, 5] > 20000
    got = s4g.scan_host(db, q, sc, block_pipe=True)
    st = s4gd.scan_stats(0)
    assert np.array_equal(got, exp)
    assert st["list_pairs"] == len(qs) * len(ts) and st["speculative"] >= 1 and 1 <= st["rechecked"] <= 4 and st["tier_drift"] == 0 and st["pair_items"] == 0, st
    got = s4g.scan_host(db, q, sc, block_pipe=True, no_spec=True)
    st = s4gd.scan_stats(0)
    assert np.array_equal(got, exp) and st["list_pairs"] == 0 and st["speculative"] == 0 and st["rechecked"] == 0, st
    one = s4g.QuerySet([qs[2]])
    got = s4g.scan_host(db, one, sc, block_pipe=True)
    st = s4gd.scan_stats(0)
    assert np.array_equal(got[0], exp[2]) and st["list_pairs"] == len(ts) and st["rechecked"] >= 1, st
    # other gap penalties: a frame that drifts twice as fast (lower mark), and one that no longer fits (the pipeline refuses: tiers by bound)
    for go, ge in ((11, 2), (12, 5)):
        sc2 = s4g.Scorer("BLOSUM_62", go, ge)
        exp2 = oracle_mod.scan(qs, ts, blosum62, go, ge, threads=8)
        assert np.array_equal(s4g.scan_host(db, q, sc2, block_pipe=True), exp2), (go, ge)


@pytest.mark.parametrize("waves", ["4", "12"])
def test_block_pipeline_with_four_and_twelve_waves_per_block(s4g, oracle_mod, blosum62, scorer, monkeypatch, waves):
    """Round 6: the pipeline's block is four waves, or twelve where a call is bound by the latency of its longest item (few long queries
    against a large database: one block to a CU, an item's strips on three waves per SIMD).  S4G_SCAN_PIPE_WAVES forces one form: every
    letter code, ragged groups, query counts from 1 to 9 (items that end inside a block's round of strips, items shorter than the
    block has waves), strips around the strip height, a long query beyond the window (speculative) -- the same matrix as the oracle's;
    and the library's own choice on a call of one long query against 70 000 targets equals the forced forms."""
    import torch
    monkeypatch.setenv("S4G_SCAN_PIPE_WAVES", waves)
    rng = np.random.default_rng(640 + int(waves))
    ts = rand_seqs(rng, 700, 1, 500, 26) + [rand_seqs(rng, 1, 3900, 3900)[0]]
    for qlens in ([1], [33, 31], [350, 97, 5], [64, 65, 63, 32, 1, 128, 129, 700, 12], [5200, 2100, 40]):
        qs = [rand_seqs(rng, 1, l, l, 26 if l < 1000 else 20)[0] for l in qlens]
        ts[7] = qs[0].copy()
        exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=os.cpu_count() or 8)
        got = s4g.scan_host(s4g.Database(ts), s4g.QuerySet(qs), scorer, block_pipe=True)
        bad = np.argwhere(got != exp)
        assert bad.size == 0, (waves, qlens, bad[:5].tolist())
    if waves == "12":                       # the library's own choice against the forced forms, at a size where it takes the pipeline by itself
        from sift4g_amd import synth
        cfg = synth.make_config(2, n_queries=2, n_targets=70000, homologs_per_query=2)
        db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])
        q = s4g.QuerySet([rand_seqs(rng, 1, 2600, 2600)[0]])
        outs = []
        for w in ("12", "4", None):
            if w: monkeypatch.setenv("S4G_SCAN_PIPE_WAVES", w)
            else: monkeypatch.delenv("S4G_SCAN_PIPE_WAVES")
            out = torch.empty((1, db.n), dtype=torch.int32, device="cuda:0")
            s4g.scan(db, q, scorer, out)
            torch.cuda.synchronize()
            from sift4g_amd import device as s4gd
            assert s4gd.scan_stats(0)["list_pairs"] == db.n
            outs.append(out)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_odd_query_counts_pair_groups_in_the_block_pipeline(s4g, oracle_mod, blosum62, scorer, monkeypatch):
    """Round 6: a query without a partner takes two neighbouring GROUPS of the tile per item (sw_scan_lists<R, 2>: the halves of a register
    read different targets) instead of computing every cell twice.  One, three and seven queries against a tile with an odd number of
    groups, a ragged last group, targets of very different lengths in neighbouring groups (the shorter group is padded in flight), every
    letter code: the same matrix as the oracle's and as the form that repeats the query (S4G_SCAN_PIPE_NO_GROUP_PAIRS), four and twelve
    waves to a block; and one query against 200 000 targets takes the pipeline by itself."""
    import torch
    rng = np.random.default_rng(650)
    ts = rand_seqs(rng, 64 * 4 + 17, 1, 700, 26)                    # five groups: the last item of a lone query has no second group
    ts[3] = rand_seqs(rng, 1, 3500, 3500)[0]
    for qlens in ([300], [61, 350, 97], [5, 33, 64, 65, 128, 200, 410]):
        qs = [rand_seqs(rng, 1, l, l, 26)[0] for l in qlens]
        ts[9] = qs[-1].copy()
        exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=os.cpu_count() or 8)
        db, q = s4g.Database(ts), s4g.QuerySet(qs)
        for waves in ("4", "12"):
            monkeypatch.setenv("S4G_SCAN_PIPE_WAVES", waves)
            got = s4g.scan_host(db, q, scorer, block_pipe=True)
            assert np.array_equal(got, exp), (qlens, waves, np.argwhere(got != exp)[:5].tolist())
            monkeypatch.setenv("S4G_SCAN_PIPE_NO_GROUP_PAIRS", "1")
            assert np.array_equal(s4g.scan_host(db, q, scorer, block_pipe=True), exp), (qlens, waves, "repeated query")
            monkeypatch.delenv("S4G_SCAN_PIPE_NO_GROUP_PAIRS")
        monkeypatch.delenv("S4G_SCAN_PIPE_WAVES")
    from sift4g_amd import synth, device as s4gd
    cfg = synth.make_config(3, n_queries=2, n_targets=200000, homologs_per_query=2)
    db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])
    q = s4g.QuerySet([rand_seqs(rng, 1, 320, 320)[0]])
    out, ref = (torch.empty((1, db.n), dtype=torch.int32, device="cuda:0") for _ in range(2))
    s4g.scan(db, q, scorer, out)
    torch.cuda.synchronize()
    assert s4gd.scan_stats(0)["list_pairs"] == db.n                      # the library's own choice: the pipeline
    s4g.scan(db, q, scorer, ref, no_pairs=True)                          # the scan kernels
    assert torch.equal(out, ref)


def test_pair_scan_bands_for_long_queries(s4g, oracle_mod, blosum62, scorer):
    """sw_pair_scan_band: queries of 1025 .. 5000 residues (2 to 5 bands of 1024 rows, partial last bands of 1 row and of
    1023 rows), alignments that cross band boundaries, targets across window boundaries, three scorers, and the scan
    entry point choosing the pair route for a long query on its own."""
    rng = np.random.default_rng(39)
    qlens = [1025, 1026, 2047, 2048, 2049, 3000, 3073, 4096, 5000]
    qs = [rand_seqs(rng, 1, l, l, 26)[0] for l in qlens]
    ts = rand_seqs(rng, 60, 1, 900, 26) + [rand_seqs(rng, 1, l, l, 26)[0] for l in (2047, 2049, 4100, 6000)]
    for k, src in enumerate(qs):                               # homologs of a stretch straddling the first band boundary
        piece = src[700:1500].copy()
        piece[rng.integers(0, len(piece), size=60)] = rng.integers(0, 20, size=60)
        ts[k] = np.concatenate([rand_seqs(rng, 1, 40, 40)[0], piece[:390], piece[400:], rand_seqs(rng, 1, 25, 25)[0]])
    ts[20] = qs[5].copy()                                      # a whole 3000-residue query as a target: path through 3 bands
    ts[21] = np.concatenate([qs[8][:2500], qs[8][2510:]])
    pq, pt = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
    for go, ge in ((10, 1), (2, 2), (30, 3)):
        sc = s4g.Scorer("BLOSUM_62", go, ge)
        _pairs_check(s4g, oracle_mod, blosum62, qs, ts, pq.ravel(), pt.ravel(), sc, go, ge, routes=("wave", None))
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    assert np.array_equal(s4g.scan_host(db, q, scorer, q_begin=5, q_count=1), exp[5:6])       # pair route, banded
    assert np.array_equal(s4g.scan_host(db, q, scorer, no_pairs=True), exp)
    # banded traceback: alignments through one, two and three band boundaries, against the oracle and the strip kernels
    MOVE = {0: 1, 1: 2, 2: 3}
    sel = [(k, k) for k in range(len(qs))] + [(5, 20), (8, 21), (8, 63), (3, 62), (0, 30), (7, 61)]
    pq2 = np.array([a for a, _ in sel]); pt2 = np.array([b for _, b in sel])
    res = s4g.traceback(db, q, scorer, pq2, pt2, exp[pq2, pt2])
    ref = s4g.traceback(db, q, scorer, pq2, pt2, exp[pq2, pt2], route="scan")
    for a, b, r, g in zip(pq2, pt2, res, ref):
        o = oracle_mod.sw_align(qs[a], ts[b], blosum62, 10, 1)
        assert o["score"] == r["score"] and r["path"].tolist() == [MOVE[m] for m in o["path"]], (int(a), int(b))
        assert (r["q_start"], r["q_end"], r["t_start"], r["t_end"]) == (o["q_start"], o["q_end"], o["t_start"], o["t_end"])
        assert g["path"].tolist() == r["path"].tolist()
    assert len(res[len(qs)]["path"]) >= 3000                  # the 3000-residue self alignment crosses two band boundaries


def test_very_long_targets_take_the_pair_route_per_tile(s4g, oracle_mod, blosum62, scorer):
    """a few titin-sized targets among ordinary ones: their tile (sequences beyond 12288 residues) is scored by one wave
    per pair while the rest of the database runs through the scan kernels, in s4g_scan and in the streaming top-k;
    same scores and the same top-k as with the scan kernels alone."""
    rng = np.random.default_rng(40)
    qs = rand_seqs(rng, 3, 150, 420, 26) + [rand_seqs(rng, 1, 1300, 1300, 26)[0]]
    ts = rand_seqs(rng, 400, 20, 700, 26) + [rand_seqs(rng, 1, l, l, 26)[0] for l in (12289, 15000, 20000)]
    ts[401][9000:9400] = qs[0][:400] if len(qs[0]) >= 400 else np.resize(qs[0], 400)
    ts[402][18000:18000 + len(qs[1])] = qs[1]
    ts[7] = qs[2][10:].copy()
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    assert len(db.tiles()) == 2
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    for mode in (dict(), dict(no_pairs=True)):
        assert np.array_equal(s4g.scan_host(db, q, scorer, **mode), exp), mode
    for k, ms in ((5, 1), (50, 40)):
        want = None
        for mode in (dict(), dict(no_pairs=True)):
            ids, sc, cnt = s4g.scan_topk(db, q, scorer, k, ms, **mode)
            got = (ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy())
            for i in range(len(qs)):
                order = sorted((j for j in range(len(ts)) if exp[i, j] >= ms), key=lambda j: (-exp[i, j], j))[:k]
                assert got[0][i, :len(order)].tolist() == order and int(got[2][i]) == len(order), (k, ms, mode, i)
                assert got[1][i, :len(order)].tolist() == [int(exp[i, j]) for j in order]


def test_long_tile_pair_route_inside_a_large_scan(s4g, oracle_mod, blosum62, scorer):
    """a call too large for the whole-call pair route (64 queries x 70k targets) whose three 13k-20k-residue targets sit in
    a tile of their own: s4g_scan scores that tile with one wave per pair on the side stream while the scan kernels do the
    rest; same matrix as with the scan kernels alone, spot-checked against the oracle (all long columns, 2000 others)."""
    rng = np.random.default_rng(41)
    qs = rand_seqs(rng, 64, 200, 420, 26)
    lens = rng.integers(20, 180, size=70000)
    ts = [rand_seqs(rng, 1, int(l), int(l), 26)[0] for l in lens] + [rand_seqs(rng, 1, l, l, 26)[0] for l in (13000, 16001, 20000)]
    ts[70001][5000:5000 + len(qs[3])] = qs[3]
    db, q = s4g.Database(ts), s4g.QuerySet(qs)
    assert len(db.tiles()) == 2 and len(qs) * len(ts) > (4 << 20)
    got = s4g.scan_host(db, q, scorer)
    ref = s4g.scan_host(db, q, scorer, no_pairs=True)
    assert np.array_equal(got, ref)
    cols = np.concatenate([np.arange(70000, 70003), rng.choice(70000, size=2000, replace=False)])
    sub = [ts[c] for c in cols]
    exp = oracle_mod.scan(qs, sub, blosum62, 10, 1, threads=8)
    assert np.array_equal(got[:, cols], exp)
    assert got[3, 70001] > 1000


def test_partial_group_split_mid_size_scan(s4g, oracle_mod, blosum62, scorer):
    """a few queries against 20k targets with a long-tailed length distribution: one or two queries -- s4g_scan gives the
    longest groups of the tile to pair waves and the rest to the scan kernels (as the streaming top-k does for any count);
    three to eight -- the whole call runs on the block-cooperative strip pipeline (sw_scan_lists in scan mode).  The matrix
    equals the scan kernels' and the oracle's, for several query counts (different split points), the pipeline also forced."""
    from sift4g_amd import device
    rng = np.random.default_rng(42)
    lens = np.clip(np.exp(rng.normal(5.6, 0.6, size=20000)), 20, 3000).astype(int)
    ts = [rand_seqs(rng, 1, int(l), int(l), 26)[0] for l in lens]
    qs = rand_seqs(rng, 6, 250, 420, 26)
    ts[11] = qs[0][20:].copy()
    db = s4g.Database(ts)
    tc, to = oracle_mod._flat(ts)
    qc, qo = oracle_mod._flat(qs)
    exp = oracle_mod.scan_simd_flat(qc, qo, tc, to, blosum62, 10, 1, threads=8)
    if exp is None:
        exp = oracle_mod.scan_flat(qc, qo, tc, to, blosum62, 10, 1, threads=8)
    q = s4g.QuerySet(qs)
    for nq in (1, 2, 3, 4, 6):
        assert np.array_equal(s4g.scan_host(db, q, scorer, q_begin=0, q_count=nq), exp[:nq]), nq
        st = device.scan_stats(0)
        assert st["list_pairs"] == (nq * len(ts) if nq >= 3 else 0), st      # 3..8 queries: the whole call on the list kernel's pipeline
        assert np.array_equal(s4g.scan_host(db, q, scorer, q_begin=0, q_count=nq, no_pairs=True), exp[:nq]), nq
        assert np.array_equal(s4g.scan_host(db, q, scorer, q_begin=6 - nq, q_count=nq, block_pipe=True), exp[6 - nq:]), nq
    # a database with a tile of titin-sized sequences: that tile goes through the general route, the rest through the pipeline
    giants = rand_seqs(rng, 3, 13000, 15000, 20)
    mixed = ts[:17000] + giants
    dbm = s4g.Database(mixed)
    exp_m = np.concatenate([exp[:4, :17000], oracle_mod.scan(qs[:4], giants, blosum62, 10, 1, threads=8)], axis=1)
    q4 = s4g.QuerySet(qs[:4])
    assert np.array_equal(s4g.scan_host(dbm, q4, scorer), exp_m)
    st = device.scan_stats(0)
    assert st["list_pairs"] == 4 * 17000 and st["pair_items"] + st["tier_drift"] + st["tier_m3"] + st["tier_i16"] + st["tier_i32"] > 0, st
    # a query whose score bound is beyond the drifting-frame window: since round 6 the pipeline takes the tile speculatively (its strips mark
    # what they cannot vouch for, sw_pair_recheck recomputes); told not to speculate it refuses the tile and the general route scores the call
    long_q = rand_seqs(rng, 1, 6000, 6000, 20) + qs[:2]
    ql = s4g.QuerySet(long_q)
    sub = ts[:4000]
    dbs = s4g.Database(sub)
    exp_l = oracle_mod.scan(long_q, sub, blosum62, 10, 1, threads=8)
    assert np.array_equal(s4g.scan_host(dbs, ql, scorer, block_pipe=True), exp_l)
    st = device.scan_stats(0)
    assert st["list_pairs"] == 3 * 4000 and st["speculative"] >= 1, st
    assert np.array_equal(s4g.scan_host(dbs, ql, scorer, block_pipe=True, no_spec=True), exp_l)
    assert device.scan_stats(0)["list_pairs"] == 0
    ids, sc, cnt = s4g.scan_topk(db, q, scorer, 30, 1)
    ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
    for i in range(len(qs)):
        order = sorted(range(len(ts)), key=lambda j: (-exp[i, j], j))[:30]
        assert ids[i].tolist() == order and sc[i].tolist() == [int(exp[i, j]) for j in order]


@pytest.mark.parametrize("mode", ["all", None, "0"])
def test_cooperative_pair_kernels(mode):
    """sw_pair_scan_coop / sw_pair_trace_coop (the four waves of a block on one pair, bands of 64 x RPL rows that run concurrently):
    scores and tracebacks equal the oracle's for every band count of every class, alignments across band boundaries, targets around
    the publication window, cheap and expensive gaps -- with every eligible pair forced there (S4G_COOP_PAIRS=all), under the
    library's own rule (a per-query call sends its longest pairs, a batch does not), and with the route off."""
    import subprocess, sys
    env = dict(os.environ)
    env.pop("S4G_COOP_PAIRS", None)
    if mode is not None:
        env["S4G_COOP_PAIRS"] = mode
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "coop_pairs_check.py")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok"), (r.stdout[-500:], r.stderr[-3000:])


def test_scan_pipeline_switch_in_a_fresh_process():
    """S4G_SCAN_PIPE_MAX_QUERIES (read once per process): 0 keeps every small call on the split model, 5 takes calls of 3..5 queries to
    the block pipeline -- the score matrices are the oracle's either way and s4g_scan_stats tells which route ran."""
    import subprocess, sys
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, oracle, sift4g_amd as s4g
from sift4g_amd import device
from conftest import rand_seqs
rng = np.random.default_rng(9)
lens = np.clip(np.exp(rng.normal(5.4, 0.6, size=17000)), 20, 2500).astype(int)
ts = [rand_seqs(rng, 1, int(l), int(l), 20)[0] for l in lens]
qs = rand_seqs(rng, 7, 200, 380, 20)
db, q = s4g.Database(ts), s4g.QuerySet(qs)
sc = s4g.Scorer("BLOSUM_62", 10, 1)
tc, to = oracle._flat(ts); qc, qo = oracle._flat(qs)
exp = oracle.scan_simd_flat(qc, qo, tc, to, oracle.blosum62(), 10, 1, threads=8)
if exp is None: exp = oracle.scan_flat(qc, qo, tc, to, oracle.blosum62(), 10, 1, threads=8)
want = int(os.environ["S4G_SCAN_PIPE_MAX_QUERIES"])
for nq in (2, 4, 7):
    assert np.array_equal(s4g.scan_host(db, q, sc, q_begin=0, q_count=nq), exp[:nq]), nq
    piped = device.scan_stats(0)["list_pairs"]
    assert piped == (nq * len(ts) if 3 <= nq <= want else 0), (nq, want, piped)
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    for limit in ("0", "5"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, S4G_SCAN_PIPE_MAX_QUERIES=limit))
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (limit, r.stdout[-500:], r.stderr[-2000:])
