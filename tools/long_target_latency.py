#!/usr/bin/env python3
"""A handful of queries against a database that contains a few titin-sized targets (what a real protein database looks
like): time of s4g_scan and of the streaming s4g_scan_topk with the library's routing (the long-sequence tile goes to one
wave per pair) and with the scan kernels alone.  usage: tools/long_target_latency.py [n_queries=5] [n_targets=100000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift4g_amd as s4g
from sift4g_amd import synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
c = synth.make_config(2, nq, nt)
qs = [c["q_codes"][c["q_off"][i]:c["q_off"][i + 1]] for i in range(nq)]
ts = [c["t_codes"][c["t_off"][i]:c["t_off"][i + 1]] for i in range(nt)]
rng = np.random.default_rng(3)
ts += [rng.integers(0, 20, size=l).astype(np.uint8) for l in (13000, 21000, 35000)]
db, q = s4g.Database(ts), s4g.QuerySet(qs)
scorer = s4g.Scorer("BLOSUM_62", 10, 1)
cells = sum(len(x) for x in qs) * sum(len(x) for x in ts)
out = torch.empty((nq, len(ts)), dtype=torch.int32, device="cuda:0")
res = {}
for name, kw in (("library routing", {}), ("scan kernels only", dict(no_pairs=True))):
    for fn in ("scan", "scan_topk"):
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if fn == "scan": s4g.scan(db, q, scorer, out, **kw)
            else: r = s4g.scan_topk(db, q, scorer, 400, 1, **kw)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res[(name, fn)] = out.clone() if fn == "scan" else r[1].clone()
        print(f"{nq} queries x {len(ts)} targets (tiles {db.tiles()}), {fn:9s}, {name:17s}: {best * 1e3:8.2f} ms  {cells / best / 1e9:8.1f} GCUPS")
assert torch.equal(res[("library routing", "scan")], res[("scan kernels only", "scan")])
assert torch.equal(res[("library routing", "scan_topk")], res[("scan kernels only", "scan_topk")])
