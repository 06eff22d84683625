#!/usr/bin/env python3
"""configs[3]-style run at a size one MI355X holds: a database of N synthetic targets GENERATED ON THE DEVICE (nothing but
the offsets ever visits the host), packed where it lies (s4g_db_create_from_device), then the streaming scan with running
top-k (s4g_scan_topk: no Q x N score matrix exists anywhere).  With the default N the database has more than 2^32
residues, so this is also the test of every 64-bit offset on the path.

Checks (size-independent properties; the oracle cannot visit 10^14 cells):
  * every reported (query, target, score) is re-scored through the one-wave-per-pair kernels, an independent implementation;
  * each query has one planted full-length copy of itself in the database: it must be rank 0 with the query's self-score;
  * completeness: random (query, target) pairs scored on their own must either be in the list or not beat its last entry.

usage: tools/big_db_scan.py [n_targets=20000000] [n_queries=64] [k=400]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift4g_amd as s4g                                   # noqa: E402
from sift4g_amd import device as s4gd                      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dev = torch.device("cuda", 0)
torch.manual_seed(0x51F74603)
rng = np.random.default_rng(0x51F74603)
LETTERS = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8) - 65

t0 = time.time()
lengths = torch.exp(torch.randn(N, device=dev) * 0.55 + 5.68).clamp_(30, 3000).to(torch.int64)
offsets = torch.zeros(N + 1, dtype=torch.int64, device=dev)
torch.cumsum(lengths, 0, out=offsets[1:])
total = int(offsets[-1])
table = torch.from_numpy(LETTERS.copy()).to(dev)
codes = torch.empty(total, dtype=torch.uint8, device=dev)
CH = 1 << 28
for a in range(0, total, CH):
    b = min(total, a + CH)
    codes[a:b] = table[torch.randint(0, 20, (b - a,), device=dev, dtype=torch.int32).long()]
torch.cuda.synchronize()
print(f"generated {N} targets, {total} residues ({total / 2**32:.2f} x 2^32) on the device in {time.time() - t0:.1f} s", flush=True)

# queries on the host; one full copy of each planted at the head of a target that is long enough
q_len = rng.integers(300, 401, Q)
q_codes = [LETTERS[rng.integers(0, 20, int(l))] for l in q_len]
h_len = lengths.cpu().numpy()
planted = []
for i in range(Q):
    p = (i + 1) * (N // (Q + 1))
    while h_len[p] < q_len[i] or p in planted:
        p += 1
    planted.append(p)
    o = int(offsets[p])
    codes[o:o + int(q_len[i])] = torch.from_numpy(q_codes[i].copy()).to(dev)
torch.cuda.synchronize()

t0 = time.time()
db = s4g.Database.from_device(codes, offsets)
torch.cuda.synchronize()
t_create = time.time() - t0
info = db.info()
print(f"s4g_db_create_from_device: {t_create:.2f} s, {len(db.tiles())} tiles, {info['hbm_bytes'] / 2**30:.1f} GiB resident, "
      f"padding {info['padded_residues'] / info['residues'] - 1:.3%}", flush=True)
del codes
torch.cuda.empty_cache()

scorer = s4g.Scorer("BLOSUM_62", 10, 1)
q_off = np.concatenate([[0], np.cumsum(q_len)]).astype(np.uint64)
qs = s4g.QuerySet(codes=np.concatenate(q_codes).astype(np.uint8), offsets=q_off)
cells = float(q_len.sum()) * float(total)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    ids, sc, cnt = s4g.scan_topk(db, qs, scorer, K)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"scan + running top-{K}: {dt:.2f} s, {cells:.3e} cells, {cells / dt / 1e9:.0f} GCUPS"
          f" (plan: {s4gd.scan_plan(db, qs, scorer)['kernel']})", flush=True)
ids, sc, cnt = ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()
assert (cnt == K).all(), cnt

# (1) planted copies: rank 0 with the self-score
M = s4gd.matrix_by_name("BLOSUM_62")
for i in range(Q):
    self_score = int(M[q_codes[i], q_codes[i]].sum())
    assert ids[i, 0] == planted[i] and sc[i, 0] == self_score, (i, ids[i, 0], planted[i], sc[i, 0], self_score)
# (2) every reported hit re-scored by the pair kernels
pq = np.repeat(np.arange(Q, dtype=np.uint32), K)
pt = ids.reshape(-1).astype(np.uint32)
again = s4g.score_pairs(db, qs, scorer, pq, pt, route="wave")
assert (again == sc.reshape(-1)).all(), int((again != sc.reshape(-1)).sum())
# order: score descending, ties by id ascending
for i in range(Q):
    key = list(zip(-sc[i].astype(np.int64), ids[i].astype(np.int64)))
    assert key == sorted(key), i
# (3) completeness on a random sample
S = 400_000
rq = rng.integers(0, Q, S).astype(np.uint32)
rt = rng.integers(0, N, S).astype(np.uint32)
rs = s4g.score_pairs(db, qs, scorer, rq, rt, route="wave")
last = sc[:, K - 1]
last_id = ids[:, K - 1]
member = {(int(q), int(t)) for q, t in zip(pq, pt)}
viol = 0
for q, t, s in zip(rq, rt, rs):
    if (int(q), int(t)) in member:
        continue
    if s > last[q] or (s == last[q] and t < last_id[q]):
        viol += 1
assert viol == 0, viol
print(f"checks passed: {Q} planted copies at rank 0 with their self-scores; {Q * K} reported hits and {S} random pairs "
      f"re-scored by the one-wave-per-pair kernels (bit-equal; no sampled pair beats a list's last entry)", flush=True)
