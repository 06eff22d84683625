"""The k-mer prefilter on the device (s4g_prefilter_*) against oracle/kmer_prefilter.c -- ids, counts and
float scores bit for bit -- and against the lists the reference's own code printed (tests/golden/prefilter.json)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

pytestmark = pytest.mark.gpu
LETTERS = "ARNDCQEGHILKMFPSTWYV"


def _both(queries, chunks, k, max_candidates):
    """the same chunks through the device and through the oracle"""
    import oracle
    from sift4g_amd import device as s4gd
    dev = s4gd.Prefilter(queries, k, max_candidates)
    ref = oracle.Prefilter(queries, k, max_candidates)
    for c in chunks:
        dev.add(c)
        ref.add(c)
    d_ids, d_counts, d_scores = dev.finish()
    r_ids, r_counts, r_scores, cut = ref.finish()
    stats = dev.stats()
    dev.close()
    return (d_ids, d_counts, d_scores), (r_ids, r_counts, r_scores, cut), stats


def _assert_equal(dev, ref):
    d_ids, d_counts, d_scores = dev
    r_ids, r_counts, r_scores, _ = ref
    assert d_counts.tolist() == r_counts.tolist()
    for q in range(len(d_counts)):
        n = int(d_counts[q])
        assert d_ids[q, :n].tolist() == r_ids[q, :n].tolist(), q
        assert d_scores[q, :n].view(np.uint32).tolist() == r_scores[q, :n].view(np.uint32).tolist(), q


def _workload(seed, nq, nt, homologs=8, qlen=(30, 400), tlen=(3, 600)):
    rng = np.random.default_rng(seed)
    queries = [rng.integers(0, 20, size=int(n)).astype(np.uint8) for n in rng.integers(*qlen, size=nq)]
    targets = [rng.integers(0, 20, size=int(n)).astype(np.uint8) for n in rng.integers(*tlen, size=nt)]
    for q, s in enumerate(queries):
        for h in range(homologs):
            m = s.copy()
            idx = rng.integers(0, len(m), size=int(len(m) * 0.05 * (h + 1)))
            m[idx] = rng.integers(0, 20, size=len(idx))
            if h % 3 == 1 and len(m) > 30:
                c = int(rng.integers(5, len(m) - 5))
                m = np.concatenate([m[:c], rng.integers(0, 20, size=int(rng.integers(1, 7))).astype(np.uint8), m[c + 2:]])
            targets[int(rng.integers(0, nt))] = np.concatenate([rng.integers(0, 20, size=int(rng.integers(0, 40))).astype(np.uint8), m])
    return queries, targets


def test_golden_lists_of_the_reference_code():
    import make_prefilter_golden as g
    import oracle
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "prefilter.json")))
    for c in cases:
        qs, ts = g.make_case(c["seed"], c["nq"], c["nt"], c["repeats"])
        queries, targets = [oracle.encode(q) for q in qs], [oracle.encode(t) for t in ts]
        dev, ref, _ = _both(queries, [targets], c["k"], c["max_candidates"])
        _assert_equal(dev, ref)
        if c.get("digest"):
            got = [[int(x) for x in dev[0][q, :dev[1][q]]] for q in range(c["nq"])]
            assert g.lists_digest(got) == c["lists_sha256"], c["name"]
            continue
        for q, want in enumerate(c["lists"]):
            got = [int(x) for x in dev[0][q, :dev[1][q]]]
            if "exact" in want:
                assert got == want["exact"], (c["name"], q)
            else:
                assert len(got) == want["size"] and set(want["must"]) <= set(got), (c["name"], q)


@pytest.mark.parametrize("k", [3, 4, 5])
def test_random_database_with_homologs(k):
    queries, targets = _workload(100 + k, 48, 6000 if k > 3 else 2500)
    dev, ref, stats = _both(queries, [targets], k, 5000)           # nothing cut off
    _assert_equal(dev, ref)
    assert stats["pairs"] == int(ref[1].sum()) and stats["hits"] >= stats["pairs"] and stats["long_pairs"] > 0


def test_the_cut_binds_and_ties_go_to_the_lower_index():
    queries, targets = _workload(7, 32, 8000, tlen=(20, 60))       # short targets: many equal scores 1 / len
    for k, max_candidates in ((4, 40), (5, 8)):
        dev, ref, _ = _both(queries, [targets], k, max_candidates)
        r_ids, r_counts, r_scores, cut = ref
        assert (cut > 0).all()                                         # the cut binds for every query ...
        assert sum(int((r_scores[q, :r_counts[q]] == cut[q]).sum()) for q in range(32)) >= 32     # ... and there are ties on it
        _assert_equal(dev, ref)


def test_chunks_give_the_same_lists_as_the_whole_database():
    queries, targets = _workload(8, 24, 5000)
    whole, ref, _ = _both(queries, [targets], 5, 60)
    parts, _, stats = _both(queries, [targets[:1], targets[1:1700], [], targets[1700:1701], targets[1701:]], 5, 60)
    _assert_equal(whole, ref)
    _assert_equal(parts, ref)
    assert stats["chunks"] == 5


def test_several_passes_over_one_chunk(monkeypatch):
    monkeypatch.setenv("S4G_PREFILTER_PASS_HITS", "20000")
    queries, targets = _workload(9, 40, 4000)
    dev, ref, stats = _both(queries, [targets], 4, 100)
    assert stats["hits"] > 100000                                  # at least five passes
    _assert_equal(dev, ref)


def test_long_segments_lds_and_global_tails():
    """a query of 6000 residues against itself and mutated copies (LIS beyond the 4096 tails a wave keeps in LDS), alternating
    repeats (hit lists of tens of thousands of positions, LIS bounded by the query), and ordinary 300-residue homologs"""
    rng = np.random.default_rng(5)
    big = rng.integers(0, 20, size=6000).astype(np.uint8)
    qa = np.array([16, 0] * 400, dtype=np.uint8)                   # QAQA...
    mid = rng.integers(0, 20, size=300).astype(np.uint8)
    queries = [big, qa, mid, rng.integers(0, 20, size=4500).astype(np.uint8)]
    targets = [rng.integers(0, 20, size=int(n)).astype(np.uint8) for n in rng.integers(10, 400, size=200)]
    mut = big.copy(); idx = rng.integers(0, 6000, size=300); mut[idx] = rng.integers(0, 20, size=300)
    targets[3] = big.copy()
    targets[10] = mut
    targets[11] = np.concatenate([big[3000:], big[:3000]])          # two blocks in the wrong order
    targets[20] = np.array([0, 16] * 600, dtype=np.uint8)
    targets[21] = qa.copy()
    targets[30] = np.concatenate([mid, mid, mid])
    targets[31] = queries[3][::-1].copy()
    targets[32] = queries[3].copy()
    dev, ref, stats = _both(queries, [targets], 5, 50)
    _assert_equal(dev, ref)
    assert stats["long_pairs"] >= 6
    d_ids, d_counts, d_scores = dev
    assert 3 in d_ids[0, :d_counts[0]].tolist() and d_scores[0, d_ids[0, :d_counts[0]].tolist().index(3)] == np.float32(5996) / np.float32(6000)


def test_global_tails_on_dense_random_hits():
    """two queries of 6000 residues over a four-letter alphabet, k = 3: every k-mer of a target hits ~90 query positions, so a
    (query, target) pair has hundreds of thousands of hits in no particular order and its LIS runs past the 4096 tails a wave
    keeps in LDS -- the 64-way search of pf_lis_long over tails in global memory, against the restatement's bisection"""
    rng = np.random.default_rng(77)
    queries = [rng.integers(0, 4, size=6000).astype(np.uint8), rng.integers(0, 4, size=5000).astype(np.uint8)]
    targets = [rng.integers(0, 4, size=int(n)).astype(np.uint8) for n in (6000, 3000, 4500, 5999, 64, 5000)]
    targets[3] = np.concatenate([queries[0][2000:], queries[0][:1999]])          # a rotated copy: two long increasing blocks
    dev, ref, stats = _both(queries, [targets], 3, 10)
    _assert_equal(dev, ref)
    assert stats["hits"] > 3_000_000 and stats["long_pairs"] == 12


def test_edge_cases():
    import oracle
    enc = oracle.encode
    # queries shorter than k, empty queries, empty and short targets, a database without any hit
    queries = [enc("ACD"), enc(""), enc("ACDEFGHIKL"), enc("WWWWWWWW")]
    targets = [enc(""), enc("AC"), enc("ACDEFGHIKL"), enc("YYYYYYYYYYYY"), enc("CDEFG"), enc("WWWWWWWWWWWWWWWW")]
    dev, ref, _ = _both(queries, [targets], 5, 3)
    _assert_equal(dev, ref)
    assert dev[1].tolist() == [0, 0, 2, 1]
    dev, ref, _ = _both(queries, [[enc("YYYYYYYY")], []], 5, 3)
    _assert_equal(dev, ref)
    assert dev[1].tolist() == [0, 0, 0, 0]
    # max_candidates 1; finish() twice and adding in between
    from sift4g_amd import device as s4gd
    p = s4gd.Prefilter(queries, 5, 1)
    p.add(targets[:3])
    a = p.finish()
    p.add(targets[3:])
    b = p.finish()
    assert a[1].tolist() == [0, 0, 1, 0] and b[1].tolist() == [0, 0, 1, 1] and b[0][3, 0] == 5
    p.close()


def test_arguments_are_checked():
    import oracle
    from sift4g_amd import device as s4gd
    q = [oracle.encode("ACDEFGHIKL")]
    for k in (0, 2, 6):
        with pytest.raises(RuntimeError, match="kmer_length"):
            s4gd.Prefilter(q, k, 10)
    with pytest.raises(RuntimeError, match="max_candidates"):
        s4gd.Prefilter(q, 5, 0)
    with pytest.raises(RuntimeError, match="out of range"):
        s4gd.Prefilter([np.array([1, 2, 3, 40, 5, 6], dtype=np.uint8)], 5, 10)
    p = s4gd.Prefilter(q, 5, 10)
    with pytest.raises(RuntimeError, match="out of range"):
        p.add([np.array([1, 2, 3, 4, 5, 77, 6], dtype=np.uint8)])
    p.add([oracle.encode("ACDEFGHIKL")], first_id=10)
    with pytest.raises(RuntimeError, match="ascending"):
        p.add([oracle.encode("ACDEFGHIKL")], first_id=5)
    p.close()


@pytest.mark.parametrize("mode", ["S4G_PREFILTER_RADIX_MERGE", "S4G_PREFILTER_DIRECT_SCATTER", "S4G_PREFILTER_UNPACKED", "S4G_PREFILTER_GLOBAL_SORT"])
def test_the_other_merge_paths_agree(monkeypatch, mode):
    """the candidates reach query order by a counting pass (pf_cand_count, pf_cand_scatter_staged) by default; the scatter without the LDS
    stage (the path from 2049 to 8192 queries), the radix sort they replaced (still the path beyond 8192 queries) and the all-global
    general passes give the same lists; so does the index without packed bucket words (what a query set of more than 2^23 residues, or
    of positions and query numbers beyond 31 bits together, falls back to).  Three chunks and lists of 60: the later chunks meet full
    lists (candidates dropped early)."""
    queries, targets = _workload(21, 40, 6000)
    chunks = [targets[:2500], targets[2500:2501], targets[2501:]]
    base, ref, stats = _both(queries, chunks, 5, 60)
    _assert_equal(base, ref)
    assert stats["fused"]
    monkeypatch.setenv(mode, "1")
    other, ref2, stats2 = _both(queries, chunks, 5, 60)
    _assert_equal(other, ref2)
    assert stats2["fused"] == (mode != "S4G_PREFILTER_GLOBAL_SORT")


def test_more_queries_than_the_counting_pass_holds():
    """9000 short queries: beyond kPfScatterMaxQ the merge falls back to the radix sort over the candidate regions"""
    rng = np.random.default_rng(22)
    queries = [rng.integers(0, 20, size=int(n)).astype(np.uint8) for n in rng.integers(5, 14, size=9000)]
    targets = [rng.integers(0, 20, size=int(n)).astype(np.uint8) for n in rng.integers(3, 300, size=800)]
    for i in range(0, 800, 7):
        targets[i] = np.concatenate([targets[i], queries[(i * 13) % 9000], queries[(i * 29) % 9000]])
    dev, ref, stats = _both(queries, [targets[:300], targets[300:]], 4, 25)
    _assert_equal(dev, ref)
    assert stats["fused"]


def test_chunk_already_on_the_device():
    import torch
    import oracle
    from sift4g_amd import device as s4gd
    from sift4g_amd.seqio import flatten
    queries, targets = _workload(12, 16, 3000)
    codes, off = flatten(targets)
    d_codes = torch.from_numpy(codes).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    p = s4gd.Prefilter(queries, 5, 30)
    p.add_device(d_codes, d_off)
    torch.cuda.synchronize()
    got = p.finish()
    p.close()
    ref = oracle.Prefilter(queries, 5, 30)
    ref.add(targets)
    _assert_equal(got, ref.finish())


def test_randomised_soak():
    """a time-boxed run of random configurations (k, sizes, alphabets that make hits rare or dense, repeats, planted homologs,
    empty and tiny sequences, chunking, pass size) against the restatement"""
    import time
    rng = np.random.default_rng(2024)
    deadline = time.time() + float(os.environ.get("S4G_PREFILTER_SOAK_SECONDS", "20"))      # longer runs: set the variable
    rounds = 0
    while time.time() < deadline or rounds < 6:
        k = int(rng.integers(3, 6))
        alphabet = int(rng.choice([2, 4, 20, 25]))
        nq, nt = int(rng.integers(1, 40)), int(rng.integers(1, 1500))
        queries = [rng.integers(0, alphabet, size=int(n)).astype(np.uint8) for n in rng.integers(0, 300, size=nq)]
        targets = [rng.integers(0, alphabet, size=int(n)).astype(np.uint8) for n in rng.integers(0, 400, size=nt)]
        for _ in range(int(rng.integers(0, 12))):                  # homologs and exact copies
            q = queries[int(rng.integers(0, nq))]
            if len(q):
                m = q.copy()
                m[rng.integers(0, len(m), size=len(m) // 10)] = rng.integers(0, alphabet, size=len(m) // 10)
                targets[int(rng.integers(0, nt))] = m
        if alphabet == 2 and nq * nt > 20000:                      # dense hits: keep the hit count in the millions
            queries, targets = queries[:6], targets[:300]
        max_candidates = int(rng.choice([1, 3, 50, 5000]))
        cuts = sorted(set(int(x) for x in rng.integers(0, len(targets) + 1, size=int(rng.integers(0, 4)))))
        chunks = [targets[a:b] for a, b in zip([0] + cuts, cuts + [len(targets)])]
        os.environ["S4G_PREFILTER_PASS_HITS"] = str(int(rng.choice([1000, 50000, 1 << 26])))
        switch = str(rng.choice(["", "", "S4G_PREFILTER_RADIX_MERGE", "S4G_PREFILTER_DIRECT_SCATTER", "S4G_PREFILTER_UNPACKED", "S4G_PREFILTER_GLOBAL_SORT"]))
        if switch:
            os.environ[switch] = "1"
        try:
            dev, ref, _ = _both(queries, chunks, k, max_candidates)
        finally:
            del os.environ["S4G_PREFILTER_PASS_HITS"]
            if switch:
                del os.environ[switch]
        _assert_equal(dev, ref)
        rounds += 1
    assert rounds >= 6
