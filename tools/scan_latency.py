#!/usr/bin/env python3
"""Time of one s4g_scan call for query counts between 1 and a few hundred (where launches are latency-bound): the library's
routing (block-cooperative strip pipeline / scan kernels / one wave per pair / a split of each tile between the last two;
S4G_SCAN_PIPE_MAX_QUERIES=0 takes the pipeline out of it) against the pipeline forced and the scan kernels alone.
usage: tools/scan_latency.py [n_targets=100000] [query counts ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift4g_amd as s4g
from sift4g_amd import synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
counts = [int(x) for x in sys.argv[2:]] or [1, 4, 16, 64, 256]
c = synth.make_config(2, max(counts), nt)
ts = [c["t_codes"][c["t_off"][i]:c["t_off"][i + 1]] for i in range(nt)]
db = s4g.Database(codes=c["t_codes"], offsets=c["t_off"], device=0)
scorer = s4g.Scorer("BLOSUM_62", 10, 1)
res = int(c["t_off"][-1])
for nq in counts:
    q = s4g.QuerySet(codes=c["q_codes"][: int(c["q_off"][nq])], offsets=c["q_off"][: nq + 1], device=0)
    cells = int(c["q_off"][nq]) * res
    out = torch.empty((nq, nt), dtype=torch.int32, device="cuda:0")
    line = f"{nq:4d} queries x {nt} targets:"
    keep = {}
    for name, kw in (("library routing", {}), ("block pipeline", dict(block_pipe=True)), ("scan kernels only", dict(no_pairs=True))):
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s4g.scan(db, q, scorer, out, **kw)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        keep[name] = out.clone()
        line += f"  {name} {best * 1e3:8.2f} ms ({cells / best / 1e9:7.0f} GCUPS)"
    assert torch.equal(keep["library routing"], keep["scan kernels only"]) and torch.equal(keep["block pipeline"], keep["scan kernels only"])
    print(line)
