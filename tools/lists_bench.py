#!/usr/bin/env python3
"""Throughput of candidate-list scoring (s4g_score_pairs_host) on the batched wrapper's shape: every query with its own list of
a few thousand candidates out of a resident pool (the reference's prefilter keeps <= 5000 per query, main.cpp:69).
Checks every score against the full query x pool scan (scan kernels, themselves oracle-checked) and reports GCUPS per route.
usage: tools/lists_bench.py [n_queries=200] [n_targets=50000] [candidates_per_query=3645] [reps=3]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sift4g_amd as s4g
from sift4g_amd import synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 3645
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
routes = sys.argv[5].split(",") if len(sys.argv) > 5 else ["wave", "lists", "auto"]
c = synth.make_config(2, nq, nt)
db = s4g.Database(codes=c["t_codes"], offsets=c["t_off"])
qs = s4g.QuerySet(codes=c["q_codes"], offsets=c["q_off"])
scorer = s4g.Scorer("BLOSUM_62", 10, 1)
rng = np.random.default_rng(5)
tl = np.diff(c["t_off"].astype(np.int64)); ql = np.diff(c["q_off"].astype(np.int64))
pt = np.concatenate([np.sort(rng.choice(nt, size=nc, replace=False)) for _ in range(nq)]).astype(np.uint32)
pq = np.repeat(np.arange(nq, dtype=np.uint32), nc)
cells = float((ql[pq] * tl[pt]).sum())
full = torch.empty((nq, nt), dtype=torch.int32, device="cuda:0")
s4g.scan(db, qs, scorer, full)
want = full.cpu().numpy()[pq, pt]
print(f"{nq} queries x {nc} candidates of {nt} targets = {len(pq)} pairs, {cells:.3e} cells; longest candidate {tl[pt].max()}")
for route in routes:
    r = None if route == "auto" else route
    s4g.score_pairs(db, qs, scorer, pq, pt, route=r)                        # warm (allocations)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        got = s4g.score_pairs(db, qs, scorer, pq, pt, route=r)
        best = min(best, time.perf_counter() - t0)
    bad = int((got != want).sum())
    print(f"route {route:6s}: {best * 1e3:8.2f} ms  {cells / best / 1e9:8.1f} GCUPS host to host   mismatches vs the full scan: {bad}")
    assert bad == 0
