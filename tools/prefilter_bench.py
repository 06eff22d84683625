#!/usr/bin/env python3
"""Times the k-mer prefilter on the device (s4g_prefilter_*) on a BASELINE-shaped workload and, on a sample, the CPU
restatement (oracle/kmer_prefilter.c, one thread) beside it.
  tools/prefilter_bench.py [n_queries=1000] [n_targets=100000] [kmer_length=5] [max_candidates=5000] [repeats=5] [cpu_targets=5000]
Prints residues/s through add() from host memory, through add_device() with the chunk already in HBM, and the hit / pair
counts; checks the device lists against the oracle on the CPU sample."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sift4g_amd import device as s4gd, synth  # noqa: E402
import oracle  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 5
max_c = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
cpu_nt = int(sys.argv[6]) if len(sys.argv) > 6 else 5000

c = synth.make_config(2, nq, nt)
q_codes, q_off, t_codes, t_off = c["q_codes"], c["q_off"], c["t_codes"], c["t_off"]
residues = int(t_off[-1])
queries = [q_codes[int(q_off[i]):int(q_off[i + 1])] for i in range(nq)]
print(f"{nq} queries ({int(q_off[-1])} residues) x {nt} targets ({residues} residues), k = {k}, max_candidates = {max_c}")

t0 = time.perf_counter()
p = s4gd.Prefilter(queries, k, max_c)
print(f"index of the queries' k-mers: {1e3 * (time.perf_counter() - t0):.1f} ms (first call: includes module load)")
p.add(codes=t_codes, offsets=t_off, first_id=0)              # warm: buffers grown
p.close()

best_host = best_dev = 1e9
d_codes = torch.from_numpy(t_codes).cuda()
d_off = torch.from_numpy(t_off.astype(np.int64)).cuda()
for _ in range(reps):
    t0 = time.perf_counter(); p = s4gd.Prefilter(queries, k, max_c); t_index = time.perf_counter() - t0
    t0 = time.perf_counter(); p.add(codes=t_codes, offsets=t_off, first_id=0); best_host = min(best_host, time.perf_counter() - t0)
    p.close()
    p = s4gd.Prefilter(queries, k, max_c)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); p.add_device(d_codes, d_off, first_id=0); best_dev = min(best_dev, time.perf_counter() - t0)
    t0 = time.perf_counter(); ids, counts, scores = p.finish(); t_finish = time.perf_counter() - t0
    stats = p.stats()
    p.close()
print(f"index {1e3 * t_index:.1f} ms, add from host {1e3 * best_host:.1f} ms ({residues / best_host / 1e9:.2f} G residues/s), "
      f"add from HBM {1e3 * best_dev:.1f} ms ({residues / best_dev / 1e9:.2f} G residues/s), finish {1e3 * t_finish:.1f} ms")
print(f"hits {stats['hits']}, (query, target) pairs with a hit {stats['pairs']}, of those on a whole wave {stats['long_pairs']}; "
      f"candidates kept {int(counts.sum())}")

# CPU restatement on a sample of the targets, and the check
sample = [t_codes[int(t_off[i]):int(t_off[i + 1])] for i in range(cpu_nt)]
sample_res = int(t_off[cpu_nt])
t0 = time.perf_counter()
r = oracle.Prefilter(queries, k, max_c)
r.add(sample)
r_ids, r_counts, r_scores, _ = r.finish()
t_cpu = time.perf_counter() - t0
p = s4gd.Prefilter(queries, k, max_c)
p.add(sample)
ids, counts, scores = p.finish()
ok = (counts == r_counts).all() and all((ids[q, :counts[q]] == r_ids[q, :counts[q]]).all() and
                                         (scores[q, :counts[q]].view(np.uint32) == r_scores[q, :counts[q]].view(np.uint32)).all() for q in range(nq))
print(f"CPU restatement, 1 thread, first {cpu_nt} targets ({sample_res} residues): {t_cpu:.2f} s = {sample_res / t_cpu / 1e6:.1f} M residues/s; "
      f"device lists equal: {bool(ok)}")
sys.exit(0 if ok else 3)
