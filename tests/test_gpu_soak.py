"""Randomised soak and the heavy-tailed database check, collected by pytest (round 1 kept them as scripts the driver never ran).

* test_soak_every_route: a time-boxed slice (S4G_SOAK_SECONDS, default 45) of random configurations -- scorers, tile sizes,
  lengths from 0 to a few thousand, long queries with copies of themselves -- through every device route (scan with default
  routing / scan kernels only / the block-cooperative strip pipeline / int32 in the drifting and in the fixed frame / no speculation, streaming top-k, pair lists wave /
  scan / packed list kernel / automatic, traceback one wave
  per pair / strip kernels) against the CPU oracle.  `python tests/test_gpu_soak.py 900 7` runs it for 900 s with seed 7
  (with S4G_COOP_PAIRS=all in the environment the pair routes run on the cooperative kernels).
* test_heavy_tailed_database: 200 queries x 200 000 targets plus six titin-sized outliers (15k .. 35k residues, one holding a
  planted copy of query 0): tiles, the long-sequence tile's pair route, spot checks against the oracle, streaming top-k against
  the top-k of the full matrix.
"""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rand_seqs      # noqa: E402

pytestmark = pytest.mark.gpu


def soak(budget, seed):
    import oracle
    import sift4g_amd as s4g
    rng = np.random.default_rng(seed)
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
        for mode in (dict(), dict(no_pairs=True), dict(diag_frame=True), dict(block_pipe=True), dict(force_i32=True), dict(force_i32=True, no_drift=True), dict(no_spec=True)):
            assert np.array_equal(s4g.scan_host(db, q, sc, **mode), exp), ("scan", mode, ctx)
        k = int(rng.integers(1, min(nt, 40) + 1))               # streaming top-k: tiles (alternating when >= 3), segments, both routings
        for mode in (dict(no_pairs=True), dict()):
            ids, top, cnt = s4g.scan_topk(db, q, sc, k, **mode)
            ids, top = ids.cpu().numpy(), top.cpu().numpy()
            for i in range(nq):
                order = [t for t in sorted(range(nt), key=lambda t: (-int(exp[i, t]), t)) if exp[i, t] >= 1][:k]
                assert ids[i, :len(order)].tolist() == order and top[i, :len(order)].tolist() == [int(exp[i, t]) for t in order], ("topk", mode, ctx, i)
        n = int(rng.integers(1, 400)); pq = rng.integers(0, nq, size=n); pt = rng.integers(0, nt, size=n)
        for route in ("wave", "scan", "lists", None):
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
    return it


def test_soak_every_route():
    n = soak(float(os.environ.get("S4G_SOAK_SECONDS", "45")), int(os.environ.get("S4G_SOAK_SEED", "1")))
    assert n >= 3


def test_heavy_tailed_database():
    import torch
    import oracle
    import sift4g_amd as s4g
    from sift4g_amd import synth
    cfg = synth.make_config(2, n_queries=200, n_targets=200000)
    tl = np.diff(cfg["t_off"].astype(np.int64))
    rng = np.random.default_rng(1)
    # titin-like outliers: append 6 sequences of 15k..35k residues, one of them containing a copy of query 0
    extra_len = [15000, 20000, 25000, 30000, 34350, 35000]
    extra = [synth.residues(777 + i, l) for i, l in enumerate(extra_len)]
    q0 = cfg["q_codes"][int(cfg["q_off"][0]):int(cfg["q_off"][1])]
    extra[4][17000:17000 + len(q0)] = q0
    t_codes = np.concatenate([cfg["t_codes"][:int(cfg["t_off"][-1])]] + extra)
    t_off = np.concatenate([cfg["t_off"], cfg["t_off"][-1] + np.cumsum(extra_len).astype(np.uint64)])
    t0 = time.time(); db = s4g.Database(codes=t_codes, offsets=t_off); print("db create", round(time.time() - t0, 2), "s tiles", db.tiles()[:4], "...", len(db.tiles()))
    q = s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"]); sc = s4g.Scorer()
    out = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, q, sc, out); torch.cuda.synchronize()
    t0 = time.time(); s4g.scan(db, q, sc, out); torch.cuda.synchronize(); dt = time.time() - t0
    cells = float(np.diff(cfg["q_off"].astype(np.int64)).sum()) * float(t_off[-1])
    print("scan", round(dt, 3), "s", round(cells / dt / 1e9), "GCUPS", db.info())
    got = out.cpu().numpy()
    M = oracle.blosum62()
    pq = np.concatenate([rng.integers(0, q.n, 300), np.arange(12) % q.n, np.zeros(6, int)]).astype(np.uint32)
    pt = np.concatenate([rng.integers(0, db.n, 300), db.n - 1 - (np.arange(12) % 6), db.n - 6 + np.arange(6)]).astype(np.uint32)
    exp = oracle.pairs_flat(cfg["q_codes"], cfg["q_off"], t_codes, t_off, pq, pt, M, 10, 1, threads=32)
    assert np.array_equal(got[pq, pt], exp), "MISMATCH"
    print("spot-check vs oracle OK; query0 vs titin-like with planted copy:", got[0, db.n - 2], "self-score", int(sum(M[a, a] for a in q0)))
    ids, scs, cnt = s4g.scan_topk(db, q, sc, 50, 1)
    full = s4g.topk(db, out, 50, 1)
    assert np.array_equal(ids.cpu().numpy().astype(np.uint32), full[0]) and np.array_equal(scs.cpu().numpy(), full[1])
    print("streaming top-k == full top-k OK")


if __name__ == "__main__":
    soak(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
