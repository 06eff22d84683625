#!/usr/bin/env python3
"""Latency of the per-query candidate-list call (sift4g's alignDatabase pattern): one query, a few thousand candidates
scattered over the resident database.  Compares the two exact routes of s4g_score_pairs_host: one wave per pair
(sw_pair_scan) vs scanning the whole database and picking the columns.
usage: tools/pair_latency.py [n_targets] [n_candidates] [n_queries] [query_length (default: the configs[1] generator, 300-400)]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift4g_amd as s4g
from sift4g_amd import synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 16
c = synth.make_config(2, nq, nt)
qs = [c["q_codes"][c["q_off"][i]:c["q_off"][i + 1]] for i in range(nq)]
if len(sys.argv) > 4:                      # long queries: random residues of the given length
    qlen = int(sys.argv[4])
    qs = [np.random.default_rng(100 + i).integers(0, 20, size=qlen).astype(np.uint8) for i in range(nq)]
ts = [c["t_codes"][c["t_off"][i]:c["t_off"][i + 1]] for i in range(nt)]
scorer = s4g.Scorer("BLOSUM_62", 10, 1)
db = s4g.Database(ts)
rng = np.random.default_rng(1)
tl = np.diff(c["t_off"]).astype(np.int64)
print(f"database {nt} targets, longest {tl.max()}, mean {tl.mean():.0f}; {nc} candidates per query")
for route in ("wave", "scan", None):
    per = []
    cells = 0
    for i in range(nq):
        q1 = s4g.QuerySet([qs[i]])
        cand = np.sort(rng.choice(nt, size=nc, replace=False)).astype(np.uint32)
        if i == 0:
            cand[-1] = int(np.argmax(tl))          # the longest target is somebody's candidate
        pq = np.zeros(nc, dtype=np.uint32)
        s4g.score_pairs(db, q1, scorer, pq[:8], cand[:8], route=route)      # warm
        t0 = time.perf_counter()
        got = s4g.score_pairs(db, q1, scorer, pq, cand, route=route)
        per.append(time.perf_counter() - t0)
        cells += len(qs[i]) * int(tl[cand].sum())
        if route == "wave":
            ref = s4g.score_pairs(db, q1, scorer, pq, cand, route="scan")
            assert np.array_equal(got, ref), "routes disagree"
    per = np.array(per) * 1e3
    print(f"route {str(route):5s}: median {np.median(per):7.3f} ms  max {per.max():7.3f} ms  first(with longest target) {per[0]:7.3f} ms  "
          f"{cells / per.sum() / 1e6:8.1f} GCUPS on the candidates")
# many queries in one call: throughput of the pair kernel
q_all = s4g.QuerySet(qs)
pq = np.repeat(np.arange(nq, dtype=np.uint32), nc)
pt = rng.integers(0, nt, size=nq * nc).astype(np.uint32)
ql = np.array([len(x) for x in qs], dtype=np.int64)
cells = int((ql[pq] * tl[pt]).sum())
for route in ("wave", "scan"):
    s4g.score_pairs(db, q_all, scorer, pq[:64], pt[:64], route=route)
    t0 = time.perf_counter(); s4g.score_pairs(db, q_all, scorer, pq, pt, route=route); dt = time.perf_counter() - t0
    print(f"{nq} queries x {nc} candidates in one call, route {route}: {dt * 1e3:8.2f} ms  {cells / dt / 1e9:8.1f} GCUPS on the candidates")
