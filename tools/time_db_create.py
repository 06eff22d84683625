#!/usr/bin/env python3
"""Wall time of s4g_db_create (length sort on the host, raw upload, transposition on the device) for the bench databases.
usage: tools/time_db_create.py [n_targets ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sift4g_amd as s4g
from sift4g_amd import synth

for nt in [int(x) for x in sys.argv[1:]] or [5000, 100000, 570000]:
    c = synth.make_config(2, 4, nt)
    res = int(c["t_off"][-1])
    times = []
    for rep in range(4):
        t0 = time.perf_counter()
        db = s4g.Database(codes=c["t_codes"], offsets=c["t_off"], device=0)
        times.append(time.perf_counter() - t0)
        del db
    print(f"{nt} targets, {res / 1e6:.1f} M residues: s4g_db_create {', '.join(f'{t * 1e3:.1f}' for t in times)} ms "
          f"(best {res / min(times) / 1e9:.2f} G residues/s)")
