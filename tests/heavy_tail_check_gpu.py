#!/usr/bin/env python3
"""Test infrastructure (it checks against the CPU oracle, so it lives under tests/): 200 queries x 200 000 targets plus six
titin-sized outliers (15k .. 35k residues, one holding a planted copy of query 0) -- tiles, the long-sequence tile's pair
route, spot checks of the scan against the oracle, streaming top-k against the top-k of the full matrix.
usage: python tests/heavy_tail_check_gpu.py"""
import os
import sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sift4g_amd as s4g, oracle
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
