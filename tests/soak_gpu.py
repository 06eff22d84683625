#!/usr/bin/env python3
"""Randomised soak of every device route against the CPU oracle (test infrastructure, like tests/): scan (default routing and
scan kernels only), pair lists (wave / scan / automatic), traceback (one wave per pair / strip kernels), random scorers, tile
sizes, lengths from 0 to a few thousand.  usage: python tests/soak_gpu.py [seconds=120] [seed=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
import sift4g_amd as s4g
from conftest import rand_seqs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
B62 = oracle.blosum62()
MOVE = {0: 1, 1: 2, 2: 3}
t_end = time.time() + budget
it = 0
cells = 0
while time.time() < t_end:
    it += 1
    if rng.random() < 0.5:
        M, name = B62, "BLOSUM_62"
    else:
        M = rng.integers(-12, 16, size=(26, 26)).astype(np.int32); M = np.triu(M) + np.triu(M, 1).T; name = M
    ge = int(rng.integers(0, 6)); go = ge + int(rng.integers(0, 14))
    sc = s4g.Scorer(name, go, ge)
    nq = int(rng.integers(1, 9)); nt = int(rng.integers(1, 260))
    qmax = int(rng.choice([40, 300, 700, 1100, 2600])); tmax = int(rng.choice([30, 300, 900, 2500, 5000]))
    qs = rand_seqs(rng, nq, 0, qmax, 26); ts = rand_seqs(rng, nt, 0, tmax, 26)
    long_mode = rng.random() < 0.15         # long queries + copies of them: the speculative drift tier and its recheck pass
    if long_mode:
        nq = int(rng.integers(1, 4)); nt = int(rng.integers(8, 48)); qmax, tmax = 6000, 700
        qs = rand_seqs(rng, nq, 1800, qmax, int(rng.choice([20, 26]))); ts = rand_seqs(rng, nt, 0, tmax, 26)
        for k in range(nq):
            src = qs[k]; a = int(rng.integers(0, len(src) // 3)); b = int(rng.integers(2 * len(src) // 3, len(src) + 1))
            ts[2 * k] = src[a:b].copy()
            x = src.copy(); x[rng.integers(0, len(x), size=len(x) // 12)] = rng.integers(0, 20, size=len(x) // 12)
            ts[2 * k + 1] = x
    for k in range(0, nt, 7):                                   # related pairs
        src = qs[k % nq]
        if len(src) > 12 and len(ts[k]) > 4:
            a = int(rng.integers(0, len(src) // 2)); piece = src[a:a + len(ts[k])].copy()
            if len(piece) > 8: piece[rng.integers(0, len(piece), size=len(piece) // 8)] = rng.integers(0, 20, size=len(piece) // 8)
            ts[k][:len(piece)] = piece
    db = s4g.Database(ts, tile_seqs=int(rng.choice([0, 64, 128]))); q = s4g.QuerySet(qs)
    exp = oracle.scan(qs, ts, M, go, ge, threads=8)
    cells += sum(len(x) for x in qs) * sum(len(x) for x in ts)
    ctx = (it, go, ge, nq, nt, qmax, tmax)
    for mode in (dict(), dict(no_pairs=True), dict(force_i32=True), dict(no_spec=True)):
        assert np.array_equal(s4g.scan_host(db, q, sc, **mode), exp), ("scan", mode, ctx)
    k = int(rng.integers(1, min(nt, 40) + 1))               # streaming top-k: tiles (alternating when >= 3), segments, both routings
    for mode in (dict(no_pairs=True), dict()):
        ids, top, cnt = s4g.scan_topk(db, q, sc, k, **mode)
        ids, top = ids.cpu().numpy(), top.cpu().numpy()
        for i in range(nq):
            order = [t for t in sorted(range(nt), key=lambda t: (-int(exp[i, t]), t)) if exp[i, t] >= 1][:k]
            assert ids[i, :len(order)].tolist() == order and top[i, :len(order)].tolist() == [int(exp[i, t]) for t in order], ("topk", mode, ctx, i)
    n = int(rng.integers(1, 400)); pq = rng.integers(0, nq, size=n); pt = rng.integers(0, nt, size=n)
    for route in ("wave", "scan", None):
        assert np.array_equal(s4g.score_pairs(db, q, sc, pq, pt, route=route), exp[pq, pt]), ("pairs", route, ctx)
    sub = rng.choice(n, size=min(n, 40), replace=False)
    okq = np.array([len(qs[a]) < 65535 for a in pq[sub]])
    rw = s4g.traceback(db, q, sc, pq[sub], pt[sub], exp[pq[sub], pt[sub]], route="wave")
    rs = s4g.traceback(db, q, sc, pq[sub], pt[sub], exp[pq[sub], pt[sub]], route="scan")
    for a, b, x, y in zip(pq[sub], pt[sub], rw, rs):
        o = oracle.sw_align(qs[a], ts[b], M, go, ge)
        want = [MOVE[m] for m in o["path"]]
        assert x["path"].tolist() == want and y["path"].tolist() == want, ("trace", ctx, int(a), int(b))
        if o["score"]:
            e = (o["q_start"], o["q_end"], o["t_start"], o["t_end"])
            assert (x["q_start"], x["q_end"], x["t_start"], x["t_end"]) == e and (y["q_start"], y["q_end"], y["t_start"], y["t_end"]) == e, ("ends", ctx)
print(f"soak OK: {it} random configurations, {cells / 1e9:.1f} G cells checked on every route, {budget:.0f} s")
