"""BASELINE.json configs at their FULL sizes on one MI355X: what the oracle cannot visit (10^13 .. 10^14 cells) is checked through
size-independent properties, the oracle on samples, and independent implementations of the same result.

  configs[1]  1000 queries x 100 000 targets   -- every score of the matrix (properties), exact top-400 lists, oracle samples
  configs[2]  1000 queries x 570 000 targets   -- the Swiss-Prot-scale scan bench.py strong-scales at N > 1
  configs[3]  one GPU's share of the UniRef90-scale database: 20 M targets (6.8e9 residues, beyond 2^32) generated on the
              device, streaming scan + running top-400
  configs[4]  64 queries of 2000-5000 residues x 100 000 targets (speculative drift tier + recheck)
"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def s4g():
    import sift4g_amd
    assert sift4g_amd.device_count() >= 1
    return sift4g_amd


def _properties(got, q_len, t_len, planted, scorer_max=11):
    """score matrix `got` (torch int32 on the device, [q, n]): bounds and planted homologs"""
    assert int(got.min()) >= 0
    ql = torch.from_numpy(np.asarray(q_len, dtype=np.int64)).cuda()[:, None]
    tl = torch.from_numpy(np.asarray(t_len, dtype=np.int64)).cuda()[None, :]
    assert bool((got.long() <= scorer_max * torch.minimum(ql, tl)).all())          # a local alignment has at most min(Lq, Lt) columns
    pa = torch.tensor([a for a, _ in planted], device="cuda")
    pb = torch.tensor([b for _, b in planted], device="cuda")
    hom = got[pa, pb].float()
    bg = torch.quantile(got[:64].float().flatten()[:16_000_000], 0.999)
    assert float((hom > bg).float().mean()) > 0.95                                   # planted homologs stand out from the background


def _oracle_sample(oracle_mod, cfg, got, n_random, planted, seed):
    rng = np.random.default_rng(seed)
    nq, nt = got.shape
    pq = np.concatenate([rng.integers(0, nq, n_random), [a for a, _ in planted]]).astype(np.uint32)
    pt = np.concatenate([rng.integers(0, nt, n_random), [b for _, b in planted]]).astype(np.uint32)
    exp = oracle_mod.pairs_flat(cfg["q_codes"], cfg["q_off"], cfg["t_codes"], cfg["t_off"], pq, pt, oracle_mod.blosum62(), 10, 1,
                                threads=os.cpu_count() or 8)
    have = got[torch.from_numpy(pq.astype(np.int64)).cuda(), torch.from_numpy(pt.astype(np.int64)).cuda()].cpu().numpy()
    assert np.array_equal(have, exp)


def s4gd_cells(s4g, db, q):
    from sift4g_amd import device as s4gd
    return s4gd.scan_cells(db, q)


def test_configs1_all_1000_queries(s4g, oracle_mod):
    """BASELINE configs[1] as bench.py runs it: all 1000 queries x 100 000 targets (1.2e13 cells).  Properties over the WHOLE matrix,
    idempotence, the int32 kernel on a query range against the packed tier, the oracle on 4000 random and 400 planted pairs, and
    for 16 queries the exact top-400 list against the oracle's own scan of the whole database."""
    from sift4g_amd import synth
    cfg = synth.make_config(2, n_queries=1000, n_targets=100000)
    db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])
    q = s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    got = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, got)
    _properties(got, q.lengths, db.lengths, cfg["planted"][::25])
    again = torch.empty_like(got)
    s4g.scan(db, q, scorer, again)
    assert torch.equal(got, again)                                                   # idempotent, bit for bit
    del again
    part = torch.empty((48, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, part, q_begin=500, q_count=48, force_i32=True)           # north_star's int32 scores: the same numbers
    assert torch.equal(part, got[500:548])
    _oracle_sample(oracle_mod, cfg, got, 4000, cfg["planted"][:400], 23)
    # exact top-400 of 16 queries against the oracle's scan of the whole database (AVX2 port where the CPU has it: it is itself
    # pinned to the scalar oracle by tests/test_oracle.py)
    qs = list(range(0, 1000, 64))[:16]
    ids, sc, cnt = s4g.topk(db, got, 400, 1)
    sub_off = np.zeros(len(qs) + 1, dtype=np.uint64)
    sub_off[1:] = np.cumsum(q.lengths[qs]).astype(np.uint64)
    sub_codes = np.concatenate([cfg["q_codes"][int(cfg["q_off"][i]):int(cfg["q_off"][i + 1])] for i in qs])
    scan = oracle_mod.scan_simd_flat if oracle_mod.simd_available() else oracle_mod.scan_flat
    full = scan(np.ascontiguousarray(sub_codes), sub_off, cfg["t_codes"], cfg["t_off"], oracle_mod.blosum62(), 10, 1, os.cpu_count() or 8)
    for row, i in enumerate(qs):
        order = np.lexsort((np.arange(db.n), -full[row].astype(np.int64)))[:400]
        assert cnt[i] == 400 and ids[i].astype(np.int64).tolist() == order.tolist(), i
        assert sc[i].tolist() == full[row][order].tolist(), i


def test_configs2_swissprot_scale_scan(s4g, oracle_mod):
    """BASELINE configs[2] on one GPU: the 1000 queries x 570 000 targets (6.8e13 cells) that bench.py splits over the ranks at
    N > 1.  Properties over the whole matrix, the oracle on 3000 random + 300 planted pairs, and the streaming scan's running
    top-400 (no score matrix) against the top-400 of the matrix for every query."""
    from sift4g_amd import synth
    cfg = synth.make_config(3, n_queries=1000, n_targets=570000)
    db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])
    q = s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    got = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, got)
    _properties(got, q.lengths, db.lengths, cfg["planted"][::50])
    _oracle_sample(oracle_mod, cfg, got, 3000, cfg["planted"][:300], 29)
    ids, sc, cnt = s4g.topk_device(db, got, 400, 1)
    del got
    sids, ssc, scnt = s4g.scan_topk(db, q, scorer, 400, 1)
    assert torch.equal(ids, sids) and torch.equal(sc, ssc) and torch.equal(cnt, scnt)


def test_configs2_with_an_unclipped_length_tail(s4g, oracle_mod):
    """configs[2]'s 570 000 targets WITHOUT SURVEY 8d's clip at 3000 (synth.SWISSPROT_TAIL: the lognormal
    unclipped, 0.3 % of the sequences from a 2-10 k component, eight planted giants of 10 200 .. 35 213 residues, each holding a copy
    of a query) -- what database_alignment.cpp:36-47 would stream out of a real Swiss-Prot.  256 queries (1.8e13 cells): the sequences
    beyond 12 288 residues get a tile of their own, scored beside the main tile.  The oracle on every giant x 24 queries, on every
    planted (query, giant) pair, on 300 pairs with targets of 3000 .. 10 000 residues and 2000 random ones; properties over the
    whole matrix; the scan kernels alone (no_pairs) and the streaming top-400 agree with the default routing."""
    from sift4g_amd import synth
    from sift4g_amd import device as s4gd
    cfg = synth.make_config(3, n_queries=256, n_targets=570000, tail=synth.SWISSPROT_TAIL)
    lens = np.diff(cfg["t_off"].astype(np.int64))
    giants = cfg["giants"]
    assert len(giants) == 8 and int(lens.max()) == 35213 and int((lens > 12288).sum()) >= 6 and int((lens > 3000).sum()) > 200
    db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])
    q = s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    assert len(db.tiles()) >= 2                                   # the long sequences sit in a tile of their own
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    got = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, got)
    st = s4gd.scan_stats(0)
    _properties(got, q.lengths, db.lengths, cfg["planted"][::40])
    rng = np.random.default_rng(47)
    mid = np.nonzero((lens > 3000) & (lens <= 10000))[0]
    pq = np.concatenate([np.repeat(np.arange(24), len(giants)), [a for a, b in cfg["planted"] if b in giants],
                         rng.integers(0, q.n, 300), rng.integers(0, q.n, 2000)]).astype(np.uint32)
    pt = np.concatenate([np.tile(giants, 24), [b for a, b in cfg["planted"] if b in giants],
                         rng.choice(mid, 300), rng.integers(0, db.n, 2000)]).astype(np.uint32)
    exp = oracle_mod.pairs_flat(cfg["q_codes"], cfg["q_off"], cfg["t_codes"], cfg["t_off"], pq, pt, oracle_mod.blosum62(), 10, 1,
                                threads=os.cpu_count() or 8)
    have = got[torch.from_numpy(pq.astype(np.int64)).cuda(), torch.from_numpy(pt.astype(np.int64)).cuda()].cpu().numpy()
    assert np.array_equal(have, exp)
    M = s4gd.matrix_by_name("BLOSUM_62")
    for a, b in cfg["planted"]:
        if b in giants:                                           # the copy of the query inside the giant scores the query's self-score
            qc = cfg["q_codes"][int(cfg["q_off"][a]):int(cfg["q_off"][a + 1])]
            assert int(got[a, b]) >= int(M[qc, qc].sum())
    alone = torch.empty_like(got)
    s4g.scan(db, q, scorer, alone, no_pairs=True)                 # every tile on the scan kernels
    assert torch.equal(alone, got)
    del alone
    ids, sc, cnt = s4g.topk_device(db, got, 400, 1)
    del got
    sids, ssc, scnt = s4g.scan_topk(db, q, scorer, 400, 1)
    assert torch.equal(ids, sids) and torch.equal(sc, ssc) and torch.equal(cnt, scnt)
    assert st["tier_drift"] > 0, st


def test_configs3_streaming_scan_of_one_gpus_share(s4g, oracle_mod):
    """BASELINE configs[3], one GPU's share: 20 M targets generated on the device (6.8e9 residues: every offset beyond 32 bits),
    packed where they lie, streaming scan + running top-400 for 16 queries (3.8e13 cells, 20 tiles).  A planted copy of every query
    must lead its list with the self-score; every reported hit and 100 000 random pairs are re-scored by the one-wave-per-pair
    kernels (an independent implementation) and 600 of the hits by the CPU oracle; no sampled pair beats a list's last entry."""
    from sift4g_amd import synth
    from sift4g_amd import device as s4gd
    N, Q, K = 20_000_000, 16, 400
    dev = torch.device("cuda", 0)
    codes, offsets = synth.device_database(N, synth.SEED_BASE + 4 + 0x1000, dev)
    assert int(offsets[-1]) > 2**32
    ql = synth.query_lengths(synth.SEED_BASE + 4 + 0x100, Q)
    q_codes, q_off = synth.random_set(synth.SEED_BASE + 4 + 0x200, ql)
    lens = (offsets[1:] - offsets[:-1])
    planted = []
    for i in range(Q):
        p = (i + 1) * (N // (Q + 1))
        while int(lens[p]) < ql[i]:
            p += 1
        planted.append(p)
        o = int(offsets[p])
        codes[o:o + int(ql[i])] = torch.from_numpy(q_codes[int(q_off[i]):int(q_off[i + 1])].copy()).to(dev)
    torch.cuda.synchronize()
    db = s4g.Database.from_device(codes, offsets)
    assert len(db.tiles()) >= 19
    q = s4g.QuerySet(codes=q_codes, offsets=q_off)
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    ids, sc, cnt = s4g.scan_topk(db, q, scorer, K)
    ids, sc, cnt = ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()
    assert (cnt == K).all()
    M = s4gd.matrix_by_name("BLOSUM_62")
    for i in range(Q):
        qc = q_codes[int(q_off[i]):int(q_off[i + 1])]
        assert ids[i, 0] == planted[i] and sc[i, 0] == int(M[qc, qc].sum()), i
        key = list(zip((-sc[i].astype(np.int64)).tolist(), ids[i].astype(np.int64).tolist()))
        assert key == sorted(key), i                                                  # score descending, ties by id ascending
    pq = np.repeat(np.arange(Q, dtype=np.uint32), K)
    pt = ids.reshape(-1).astype(np.uint32)
    assert np.array_equal(s4g.score_pairs(db, q, scorer, pq, pt, route="wave"), sc.reshape(-1))
    rng = np.random.default_rng(41)
    rq = rng.integers(0, Q, 100_000).astype(np.uint32)
    rt = rng.integers(0, N, 100_000).astype(np.uint32)
    rs = s4g.score_pairs(db, q, scorer, rq, rt, route="wave")
    member = set(zip(pq.tolist(), pt.tolist()))
    last, last_id = sc[:, K - 1], ids[:, K - 1]
    for a, b, s_ in zip(rq.tolist(), rt.tolist(), rs.tolist()):
        if (a, b) not in member:
            assert s_ < last[a] or (s_ == last[a] and b > last_id[a]), (a, b, s_)
    # the oracle on 600 of the reported hits (their residues come back from the device)
    pick = rng.choice(Q * K, size=600, replace=False)
    off_h = offsets.cpu().numpy()
    seqs = [codes[int(off_h[t]):int(off_h[t + 1])].cpu().numpy() for t in pt[pick]]
    t_off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    t_off[1:] = np.cumsum([len(x) for x in seqs]).astype(np.uint64)
    exp = oracle_mod.pairs_flat(q_codes, q_off, np.concatenate(seqs), t_off, pq[pick], np.arange(600, dtype=np.uint32),
                                oracle_mod.blosum62(), 10, 1, threads=os.cpu_count() or 8)
    assert np.array_equal(exp, sc.reshape(-1)[pick])


@pytest.mark.timeout(1500)
def test_configs3_ten_thousand_queries_on_a_multi_tile_share(s4g, oracle_mod):
    """BASELINE configs[3] at its own QUERY COUNT: 10 000 queries of 300-400 residues against a device-generated
    share of 1.25 M targets -- two database tiles (2^20 + the rest) and ten query batches (1024 queries fit the 4 GB
    [batch][tile columns] buffer), 1.5e15 cells.  A planted copy of every query leads its list with the self-score; every list is
    ordered (score descending, ties by ascending id: the order dbAlignmentsMerge keeps across chunks,
    /root/reference/sift4g/src/database_alignment.cpp:97-104); the lists of a 256-query slice that straddles a batch boundary equal
    s4g_topk of that slice's materialised score matrix; 2 400 sampled hits are re-scored by the CPU oracle and 200 000 by the
    one-wave-per-pair kernels; no sampled non-member beats a list's last entry."""
    import time
    from sift4g_amd import synth
    from sift4g_amd import device as s4gd
    N, Q, K = 1_250_000, 10_000, 400
    dev = torch.device("cuda", 0)
    codes, offsets = synth.device_database(N, synth.SEED_BASE + 4 + 0x5000, dev)
    ql = synth.query_lengths(synth.SEED_BASE + 4 + 0x100, Q)
    q_codes, q_off = synth.random_set(synth.SEED_BASE + 4 + 0x200, ql)
    lens_h = (offsets[1:] - offsets[:-1]).cpu().numpy()
    off_h = offsets.cpu().numpy()
    qd = torch.from_numpy(q_codes).to(dev)
    planted = np.zeros(Q, dtype=np.int64)
    p = 0
    for i in range(Q):                                        # query i copied to the head of its own target (ascending, all distinct)
        p = max(p + 1, (i + 1) * (N // (Q + 1)))
        while lens_h[p] < ql[i]:
            p += 1
        planted[i] = p
        o = int(off_h[p])
        codes[o:o + int(ql[i])] = qd[int(q_off[i]):int(q_off[i + 1])]
    torch.cuda.synchronize()
    db = s4g.Database.from_device(codes, offsets)
    assert len(db.tiles()) >= 2
    q = s4g.QuerySet(codes=q_codes, offsets=q_off)
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    cells = s4gd.scan_cells(db, q)
    t0 = time.perf_counter()
    ids_d, sc_d, cnt_d = s4g.scan_topk(db, q, scorer, K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = s4gd.scan_stats(0)
    print(f"\nconfigs[3] at 10 000 queries: {Q} x {N} targets in {len(db.tiles())} tiles, {cells:.3e} cells in {dt:.1f} s = "
          f"{cells / dt / 1e9:.0f} GCUPS (streaming scan + running top-{K}); routing {st}")
    ids, sc, cnt = ids_d.cpu().numpy(), sc_d.cpu().numpy(), cnt_d.cpu().numpy()
    assert (cnt == K).all()
    # planted copies first, with the self-score
    M = s4gd.matrix_by_name("BLOSUM_62")
    diag = M[np.arange(26), np.arange(26)]
    self_score = np.add.reduceat(diag[q_codes].astype(np.int64), q_off[:-1].astype(np.int64))
    assert np.array_equal(ids[:, 0].astype(np.int64), planted) and np.array_equal(sc[:, 0].astype(np.int64), self_score)
    # every list ordered: score descending, ties by id ascending
    key = (-(sc.astype(np.int64)) << 32) + ids.astype(np.int64)
    assert (np.diff(key, axis=1) > 0).all()
    # a 256-query slice across the first batch boundary (queries 896..1151): top-400 of the materialised matrix
    b0, nb = 896, 256
    full = torch.empty((nb, N), dtype=torch.int32, device=dev)
    s4g.scan(db, q, scorer, full, q_begin=b0, q_count=nb)
    mi, ms, mc = s4g.topk_device(db, full, K, 1)
    assert torch.equal(mi, ids_d[b0:b0 + nb]) and torch.equal(ms, sc_d[b0:b0 + nb]) and torch.equal(mc, cnt_d[b0:b0 + nb])
    part = torch.empty((8, N), dtype=torch.int32, device=dev)
    s4g.scan(db, q, scorer, part, q_begin=b0 + 124, q_count=8, force_i32=True)      # the int32 kernel on rows either side of the boundary
    assert torch.equal(part, full[124:132])
    del full, part, mi, ms, mc
    # 200 000 hits + 200 000 random pairs through the one-wave-per-pair kernels
    rng = np.random.default_rng(47)
    hq = rng.integers(0, Q, 200_000).astype(np.uint32)
    hk = rng.integers(0, K, 200_000)
    ht = ids[hq, hk].astype(np.uint32)
    assert np.array_equal(s4g.score_pairs(db, q, scorer, hq, ht, route="wave"), sc[hq, hk])
    rq = rng.integers(0, Q, 200_000).astype(np.uint32)
    rt = rng.integers(0, N, 200_000).astype(np.uint32)
    rs = s4g.score_pairs(db, q, scorer, rq, rt, route="wave")
    last, last_id = sc[rq, K - 1], ids[rq, K - 1].astype(np.int64)
    beats = (rs > last) | ((rs == last) & (rt.astype(np.int64) < last_id))         # such a pair must be a member of its query's list
    for a, b in zip(rq[beats].tolist(), rt[beats].tolist()):
        assert b in ids[a], (a, b)
    # the CPU oracle on 2 400 of the reported hits (their residues come back from the device)
    pick = rng.choice(200_000, size=2400, replace=False)
    seqs = [codes[int(off_h[t]):int(off_h[t + 1])].cpu().numpy() for t in ht[pick]]
    t_off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    t_off[1:] = np.cumsum([len(x) for x in seqs]).astype(np.uint64)
    exp = oracle_mod.pairs_flat(q_codes, q_off, np.concatenate(seqs), t_off, hq[pick], np.arange(2400, dtype=np.uint32),
                                oracle_mod.blosum62(), 10, 1, threads=os.cpu_count() or 8)
    assert np.array_equal(exp, sc[hq, hk][pick])


def test_configs4_long_queries_full_size(s4g, oracle_mod):
    """BASELINE configs[4]: 64 queries of 2000-5000 residues x the 100 000-target database (7.7e12 cells; score bounds up to 28 000:
    inside the column frame's window for nearly all of them, beyond the anti-diagonal frame's, where most queries run the
    drifting-frame kernel speculatively and the exact int32 pass recomputes what it marks).  Routing by
    bound alone (no_spec) must give the same matrix bit for bit; the oracle on 1500 random and all planted pairs of 8 queries."""
    from sift4g_amd import synth
    from sift4g_amd import device as s4gd
    cfg = synth.make_config(5, n_queries=64, n_targets=100000, q_len=(2000, 5000))
    db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])
    q = s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    got = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, got)
    st = s4gd.scan_stats(0)
    # round 5: the column frame's window has no row term (bound + 64 + 3008 columns <= 31 743 holds up to a bound of 28 671), so nearly
    # all of these queries are in the drifting-frame tier BY BOUND; the few beyond it run there speculatively
    assert st["tier_drift"] == 64 and st["speculative"] <= 8, st
    by_bound = torch.empty_like(got)
    s4g.scan(db, q, scorer, by_bound, no_spec=True)
    assert torch.equal(got, by_bound)
    # the anti-diagonal frame (rounds 1-4) needs the speculative launch for more of them (15 of 64 here): the same matrix
    s4g.scan(db, q, scorer, by_bound, diag_frame=True)
    st = s4gd.scan_stats(0)
    assert st["speculative"] > 8 and torch.equal(got, by_bound), st
    planted = [(a, b) for a, b in cfg["planted"] if a < 8]
    _oracle_sample(oracle_mod, cfg, got, 1500, planted, 31)


def test_streaming_topk_over_several_query_batches(s4g):
    """s4g_scan_topk cuts the queries into batches when a [queries][tile columns] buffer would pass 4 GB (configs[3] has 10 000
    queries): 2 500 short queries x 600 000 short targets need two batches (1 789 + 711).  Every list must equal the top-50 of the
    full score matrix, which itself is compared with the int32 kernel on a query range."""
    from sift4g_amd import synth
    rng = np.random.default_rng(77)
    nq, nt = 2500, 600_000
    ql = rng.integers(30, 61, nq)
    tl = rng.integers(20, 61, nt)
    q_codes, q_off = synth.random_set(synth.SEED_BASE + 0x77, ql)
    t_codes, t_off = synth.random_set(synth.SEED_BASE + 0x78, tl)
    for i in range(0, nq, 25):                                            # a copy of every 25th query somewhere in the database
        t = int(rng.integers(0, nt))
        m = min(int(ql[i]), int(tl[t]))
        t_codes[int(t_off[t]):int(t_off[t]) + m] = q_codes[int(q_off[i]):int(q_off[i]) + m]
    db = s4g.Database(codes=t_codes, offsets=t_off)
    q = s4g.QuerySet(codes=q_codes, offsets=q_off)
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    full = torch.empty((nq, nt), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, full)
    part = torch.empty((64, nt), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, scorer, part, q_begin=1780, q_count=64, force_i32=True, no_drift=True)
    assert torch.equal(part, full[1780:1844])
    ids, sc, cnt = s4g.topk_device(db, full, 50, 1)
    del full, part
    sids, ssc, scnt = s4g.scan_topk(db, q, scorer, 50, 1)
    assert torch.equal(cnt, scnt) and torch.equal(sc, ssc) and torch.equal(ids, sids)
    assert int(sc[:, 0].max()) > 150


def test_configs3_prefilter_streams_one_gpus_share(s4g, oracle_mod):
    """The k-mer prefilter over one GPU's share of configs[3]: the same 20 M device-generated targets (6.8e9
    residues, offsets beyond 32 bits), handed to s4g_prefilter_add_device in parts of 700 000 targets where they lie.  The planted
    copy of every query must be in its list with the score of an exact copy; the lists must not depend on how the database was
    cut into parts; the candidates of four queries are re-scored by the CPU restatement (ids fetched back from the device) and
    must carry bit-identical float scores; and no target of a 300 000-target sample that is missing from a list beats its cut."""
    import oracle
    from sift4g_amd import synth
    from sift4g_amd import device as s4gd
    N, Q, K, MAXC = 20_000_000, 64, 5, 5000
    dev = torch.device("cuda", 0)
    codes, offsets = synth.device_database(N, synth.SEED_BASE + 4 + 0x1000, dev)
    ql = synth.query_lengths(synth.SEED_BASE + 4 + 0x100, Q)
    q_codes, q_off = synth.random_set(synth.SEED_BASE + 4 + 0x200, ql)
    queries = [q_codes[int(q_off[i]):int(q_off[i + 1])] for i in range(Q)]
    lens = (offsets[1:] - offsets[:-1])
    planted = []
    for i in range(Q):
        p = (i + 1) * (N // (Q + 1))
        while int(lens[p]) < ql[i]:
            p += 1
        planted.append(p)
        o = int(offsets[p])
        codes[o:o + int(ql[i])] = torch.from_numpy(queries[i].copy()).to(dev)
    torch.cuda.synchronize()

    def run(part):
        p = s4gd.Prefilter(queries, K, MAXC)
        for t0 in range(0, N, part):
            t1 = min(N, t0 + part)
            p.add_device(codes, offsets[t0:t1 + 1], first_id=t0)          # offsets stay absolute: the part is read where it lies
        out = p.finish()
        st = p.stats()
        p.close()
        return out, st

    (ids, counts, scores), st = run(700_000)
    assert st["chunks"] == 29 and (counts == MAXC).all() and st["pairs"] > Q * MAXC
    for i in range(Q):
        row = ids[i, :MAXC].astype(np.int64)
        assert (np.diff(row) > 0).all() and row[-1] < N                   # ascending, unique, in range
        at = int(np.searchsorted(row, planted[i]))
        assert row[at] == planted[i], i
        lt = int(lens[planted[i]])
        assert scores[i, at] >= np.float32(int(ql[i]) - K + 1) / np.float32(lt), i      # at least the copy's own k-mers, in order
    (ids2, counts2, scores2), _ = run(1_300_000)
    assert np.array_equal(ids, ids2) and np.array_equal(counts, counts2) and np.array_equal(scores.view(np.uint32), scores2.view(np.uint32))
    # the restatement on the candidates of four queries
    off_h = offsets.cpu().numpy()
    fetch = lambda t: codes[int(off_h[t]):int(off_h[t + 1])].cpu().numpy()
    for i in (0, 21, 42, 63):
        seqs = [fetch(int(t)) for t in ids[i, :MAXC]]
        ref = oracle.Prefilter([queries[i]], K, MAXC)
        lis = ref.lis(seqs)[0]
        want = lis.astype(np.float32) / np.array([len(s) for s in seqs], dtype=np.float32)
        assert np.array_equal(want.view(np.uint32), scores[i, :MAXC].view(np.uint32)), i
    # nothing outside a list beats its cut: a random sample of targets through the restatement
    rng = np.random.default_rng(43)
    sample = np.unique(rng.integers(0, N, 300_000))
    seqs = [fetch(int(t)) for t in sample[:40_000]]                           # 40 000 sequences come back to the host
    check = [0, 21, 42, 63]
    ref = oracle.Prefilter([queries[i] for i in check], K, MAXC)
    lis = ref.lis(seqs)
    ln = np.array([len(s) for s in seqs], dtype=np.float32)
    for r, i in enumerate(check):
        sc = lis[r].astype(np.float32) / ln
        cut = scores[i, :MAXC].min()
        members = set(ids[i, :MAXC].tolist())
        for j in np.nonzero(sc >= cut)[0]:
            t = int(sample[j])
            if t not in members:                                              # equal to the cut and a higher index than every kept tie
                assert sc[j] == cut and t > max(int(x) for x, s_ in zip(ids[i, :MAXC], scores[i, :MAXC]) if s_ == cut), (i, t)


@pytest.mark.parametrize("N", [24_000_000, 1_500_000], ids=["8GB-file", "0.5GB-file"])
def test_configs3_share_streamed_from_storage(s4g, tmp_path_factory, N):
    """BASELINE configs[3] from storage (missing 3): one GPU's share of a UniRef-scale database -- 24 M device-generated
    targets, 8.2e9 residues, more than 8 GB as a packed file -- saved, dropped from HBM, and scanned out of the file tile by tile
    (s4g_db_open_streamed: two tile slots, the host reads tile i + 1 while tile i runs).  The streaming top-400 of 16 queries
    equals the resident scan's, list for list; the test prints the file's read rate next to the rate the scan consumes tiles at.
    The 8 GB case SKIPS -- visibly, with the free space in the reason -- on a box without 20 GiB of scratch disk; the 0.5 GB case
    (two tiles) always runs, so a green run names which of the two sizes it covered."""
    import shutil
    import time
    from sift4g_amd import synth
    base = tmp_path_factory.mktemp("s4g_stream")
    free = shutil.disk_usage(base).free
    if N == 24_000_000 and free < 20 * 2**30:
        pytest.skip(f"the 8.6 GB packed file needs 20 GiB of scratch disk; {free / 2**30:.1f} GiB free under {base}")
    Q, K = 16, 400
    dev = torch.device("cuda", 0)
    codes, offsets = synth.device_database(N, synth.SEED_BASE + 4 + 0x3000, dev)
    ql = synth.query_lengths(synth.SEED_BASE + 4 + 0x100, Q)
    q_codes, q_off = synth.random_set(synth.SEED_BASE + 4 + 0x200, ql)
    lens = (offsets[1:] - offsets[:-1])
    for i in range(Q):                                   # a copy of every query somewhere in the share
        p = (i + 1) * (N // (Q + 1))
        while int(lens[p]) < ql[i]:
            p += 1
        o = int(offsets[p])
        codes[o:o + int(ql[i])] = torch.from_numpy(q_codes[int(q_off[i]):int(q_off[i + 1])].copy()).to(dev)
    torch.cuda.synchronize()
    db = s4g.Database.from_device(codes, offsets)
    del codes, offsets
    q = s4g.QuerySet(codes=q_codes, offsets=q_off)
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    want = [x.cpu().numpy() for x in s4g.scan_topk(db, q, scorer, K)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    [x.cpu() for x in s4g.scan_topk(db, q, scorer, K)]
    t_resident = time.perf_counter() - t0
    path = base / "share.s4g"
    t0 = time.perf_counter()
    db.save(path)
    t_save = time.perf_counter() - t0
    n_tiles, cells = len(db.tiles()), s4gd_cells(s4g, db, q)
    db.close()
    torch.cuda.empty_cache()
    size = os.path.getsize(path)
    assert size > (8 * 2**30 if N == 24_000_000 else 2**28)
    sdb = s4g.Database.open_streamed(path)
    assert len(sdb.tiles()) == n_tiles
    t0 = time.perf_counter()
    got = [x.cpu().numpy() for x in s4g.scan_topk(sdb, q, scorer, K)]
    t_stream = time.perf_counter() - t0
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    st = sdb.stream_stats()
    print(f"\nconfigs[3] share from storage: {N} targets, file {size / 1e9:.2f} GB in {n_tiles} tiles (written in {t_save:.1f} s); "
          f"{Q} queries: resident scan {t_resident:.2f} s = {cells / t_resident / 1e9:.0f} GCUPS, streamed from the file {t_stream:.2f} s = "
          f"{cells / t_stream / 1e9:.0f} GCUPS; reads {st['bytes_read'] / 1e9:.2f} GB at {st['bytes_read'] / max(st['read_s'], 1e-9) / 1e9:.2f} GB/s, "
          f"the scan consumes tiles at {size / t_resident / 1e9:.2f} GB/s; the reader waited {st['wait_s']:.2f} s for the device")
    sdb.close()
    os.remove(path)
