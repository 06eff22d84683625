#!/usr/bin/env python3
"""configs[1] with the host on both sides of the boundary: everything starts and ends in host memory (the C ABI's *_host entry
points).  bench.py's `value` has the inputs resident in HBM; this is the PCIe-inclusive rate DESIGN.md quotes beside it.
usage: tools/pcie_inclusive.py [n_queries=1000] [n_targets=100000]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sift4g_amd as s4g
from sift4g_amd import synth
from sift4g_amd.device import scan_cells

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
cfg = synth.make_config(2, n_queries=nq, n_targets=nt)
scorer = s4g.Scorer("BLOSUM_62", 10, 1)
torch.cuda.init()
for rep in range(3):
    t0 = time.perf_counter()
    db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])          # H2D of the residues + packing on the device
    qs = s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    t1 = time.perf_counter()
    scores = s4g.scan_host(db, qs, scorer)                                  # scan + D2H of the whole score matrix
    t2 = time.perf_counter()
    ids, sc, cnt = s4g.scan_topk(db, qs, scorer, 400)                       # scan + top-400 on the device, D2H of the lists only
    ids, sc, cnt = ids.cpu(), sc.cpu(), cnt.cpu()
    t3 = time.perf_counter()
    cells = scan_cells(db, qs)
    print(f"rep {rep}: database + queries to HBM {1e3 * (t1 - t0):.1f} ms; scan + {scores.nbytes / 1e6:.0f} MB of scores to the host "
          f"{1e3 * (t2 - t1):.1f} ms = {cells / (t2 - t0) / 1e9:.0f} GCUPS host to host; scan + top-400 lists to the host "
          f"{1e3 * (t3 - t2):.1f} ms = {cells / (t3 - t2 + t1 - t0) / 1e9:.0f} GCUPS host to host", flush=True)
    del db, qs
