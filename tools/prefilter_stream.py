#!/usr/bin/env python3
"""The k-mer prefilter streaming one GPU's share of BASELINE configs[3]: N device-generated targets handed to
s4g_prefilter_add_device in parts where they lie.  usage: tools/prefilter_stream.py [n_targets=20000000] [n_queries=1000] [part=700000]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sift4g_amd import synth, device as s4gd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
part = int(sys.argv[3]) if len(sys.argv) > 3 else 700_000
dev = torch.device("cuda", 0)
codes, offsets = synth.device_database(N, synth.SEED_BASE + 4 + 0x1000, dev)
ql = synth.query_lengths(synth.SEED_BASE + 4 + 0x100, Q)
q_codes, q_off = synth.random_set(synth.SEED_BASE + 4 + 0x200, ql)
queries = [q_codes[int(q_off[i]):int(q_off[i + 1])] for i in range(Q)]
residues = int(offsets[-1])
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    p = s4gd.Prefilter(queries, 5, 5000)
    for a in range(0, N, part):
        b = min(N, a + part)
        p.add_device(codes, offsets[a:b + 1], first_id=a)
    t1 = time.perf_counter()
    ids, counts, scores = p.finish()
    t2 = time.perf_counter()
    st = p.stats()
    p.close()
    print(f"{Q} queries x {N} targets ({residues / 1e9:.2f} G residues) in parts of {part}: {st['chunks']} parts, {st['hits'] / 1e6:.0f} M hits, "
          f"{st['pairs'] / 1e6:.0f} M pairs; add {t1 - t0:.2f} s = {residues / (t1 - t0) / 1e9:.2f} G residues/s, finish {1e3 * (t2 - t1):.0f} ms", flush=True)
