"""Run in a fresh process (tests/test_gpu_parity.py, test_cooperative_pair_kernels): scores and tracebacks of the cooperative pair
kernels (sw_pair_scan_coop / sw_pair_trace_coop: the four waves of a block on one pair) against the oracle.  S4G_COOP_PAIRS=all in
the environment sends every pair of two bands or more there; without it the library's own rule decides (the last part)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
import sift4g_amd as s4g
from sift4g_amd import device
from conftest import rand_seqs

M = oracle.blosum62()
MOVE = {0: 1, 1: 2, 2: 3}
rng = np.random.default_rng(77)
forced = os.environ.get("S4G_COOP_PAIRS") == "all"

# every band count of every class (RPL 2: 129..512 rows, 3: 513..768, 4: 769..1024), ragged last bands, one-row last band
qlens = [129, 130, 200, 256, 257, 350, 384, 385, 511, 512, 513, 600, 641, 768, 769, 900, 1023, 1024, 100, 1100]
qs = [rand_seqs(rng, 1, l, l)[0] for l in qlens]
ts = []
for src in qs:                                  # gapped homologs inside longer targets: alignments that cross band boundaries
    cut = len(src) // 2
    ts.append(np.concatenate([rand_seqs(rng, 1, 300, 300)[0], src[:cut], rand_seqs(rng, 1, 9, 9)[0], src[cut:], rand_seqs(rng, 1, 100, 100)[0]]))
    ts.append(np.concatenate([src[:cut], src[cut + 4:], rand_seqs(rng, 1, 70, 70)[0]]))
ts += rand_seqs(rng, 12, 1, 700)
ts += [rand_seqs(rng, 1, 1, 1)[0], rand_seqs(rng, 1, 63, 63)[0], rand_seqs(rng, 1, 64, 64)[0], rand_seqs(rng, 1, 127, 127)[0], rand_seqs(rng, 1, 128, 128)[0],
       rand_seqs(rng, 1, 129, 129)[0], rand_seqs(rng, 1, 255, 255)[0], rand_seqs(rng, 1, 257, 257)[0]]      # around the window of 128 steps
long_t = rand_seqs(rng, 1, 5000, 5000)[0]
long_t[2400:2750] = qs[5]
long_t[4100:4612] = qs[9]
ts.append(long_t)
db, q = s4g.Database(ts), s4g.QuerySet(qs)
pq, pt = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
pq, pt = pq.ravel().astype(np.uint32), pt.ravel().astype(np.uint32)
for go, ge in ((10, 1), (1, 1), (0, 0), (5, 5), (40, 3)):
    sc = s4g.Scorer("BLOSUM_62", go, ge)
    exp = oracle.scan(qs, ts, M, go, ge, threads=8)
    got = s4g.score_pairs(db, q, sc, pq, pt, route="wave")
    st = device.scan_stats(0)
    assert np.array_equal(got, exp[pq, pt]), (go, ge, np.argwhere(got != exp[pq, pt])[:5].tolist())
    if forced:
        assert st["coop_scan"] == sum(1 for a, b in zip(pq, pt) if 128 < len(qs[a]) <= 1024 and len(ts[b]) > 0), st
    sub = rng.choice(len(pq), size=len(pq) if (go, ge) == (10, 1) else 200, replace=False)
    res = s4g.traceback(db, q, sc, pq[sub], pt[sub], got[sub])
    st = device.scan_stats(0)
    if forced:
        assert st["coop_trace"] > 0, st
    for a, b, r in zip(pq[sub], pt[sub], res):
        o = oracle.sw_align(qs[a], ts[b], M, go, ge)
        assert o["score"] == r["score"], (go, ge, a, b)
        if o["score"] == 0:
            assert len(r["path"]) == 0 and r["q_start"] == -1
            continue
        assert (r["q_start"], r["q_end"], r["t_start"], r["t_end"]) == (o["q_start"], o["q_end"], o["t_start"], o["t_end"]), (go, ge, a, b)
        assert r["path"].tolist() == [MOVE[m] for m in o["path"]], (go, ge, a, b)

# the library's own rule: the traceback of a giant among one query's hits (the per-query alignDatabase call) runs on the cooperative
# kernel; scores stay on one wave per pair
rng = np.random.default_rng(78)
lens = np.clip(np.exp(rng.normal(5.68, 0.55, size=3000)), 30, 3000).astype(int)
ts2 = [rand_seqs(rng, 1, int(l), int(l), 20)[0] for l in lens]
qs2 = rand_seqs(rng, 40, 300, 400, 20)
giant = rand_seqs(rng, 1, 12000, 12000, 20)[0]
giant[7000:7000 + len(qs2[0])] = qs2[0]
giant[7100:7103] = rand_seqs(rng, 1, 3, 3, 20)[0]
ts2.append(giant)
db2, q2 = s4g.Database(ts2), s4g.QuerySet(qs2)
sc = s4g.Scorer("BLOSUM_62", 10, 1)
exp2 = oracle.scan(qs2, ts2, M, 10, 1, threads=8)
one_q = np.zeros(len(ts2), dtype=np.uint32); all_t = np.arange(len(ts2), dtype=np.uint32)
got = s4g.score_pairs(db2, q2, sc, one_q, all_t, route="wave")
st1 = device.scan_stats(0)
assert np.array_equal(got, exp2[0])
top = np.argsort(-got.astype(np.int64), kind="stable")[:400].astype(np.uint32)
assert len(ts2) - 1 in top
res = s4g.traceback(db2, q2, sc, np.zeros(len(top), dtype=np.uint32), top, got[top])
st1t = device.scan_stats(0)
for b, r in zip(top, res):
    o = oracle.sw_align(qs2[0], ts2[b], M, 10, 1)
    assert o["score"] == r["score"] and r["path"].tolist() == [MOVE[m] for m in o["path"]], b
pq3 = np.repeat(np.arange(40, dtype=np.uint32), len(ts2)); pt3 = np.tile(all_t, 40)
got3 = s4g.score_pairs(db2, q2, sc, pq3, pt3, route="wave")
st40 = device.scan_stats(0)
assert np.array_equal(got3, exp2.reshape(-1))
mode = os.environ.get("S4G_COOP_PAIRS")
if mode is None:
    assert st1["coop_scan"] == 0 and st1t["coop_trace"] == 1 and st40["coop_scan"] == 0, (st1, st1t, st40)
if mode == "0":
    assert st1["coop_scan"] == 0 and st1t["coop_trace"] == 0
print("ok", st1["coop_scan"], st1t["coop_trace"], st40["coop_scan"])
