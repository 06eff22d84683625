#!/usr/bin/env python3
"""Where the time of s4g_prefilter_finish goes (1000 queries x 5000 candidates: 20 MB of ids, 20 MB of scores): the call with fresh
(never touched) output arrays, with arrays the caller has used before, ids only (what the drop-in asks for), and the library's own
breakdown (S4G_PROFILE).  usage: tools/prefilter_finish_probe.py [n_queries=1000] [n_targets=100000]"""
import os, sys, time
os.environ["S4G_PROFILE"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sift4g_amd import synth
import sift4g_amd.device as s4gd
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
c = synth.make_config(2, nq, nt)
queries = [c["q_codes"][int(c["q_off"][i]):int(c["q_off"][i + 1])] for i in range(nq)]
d_codes = torch.from_numpy(c["t_codes"]).cuda(); d_off = torch.from_numpy(c["t_off"].astype(np.int64)).cuda()
p = s4gd.Prefilter(queries, 5, 5000)
p.add_device(d_codes, d_off, first_id=0)
torch.cuda.synchronize()
def t(f, n=5):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print(f"fresh arrays, ids + scores: {t(lambda: p.finish()):.2f} ms")
print(f"fresh arrays, ids only:     {t(lambda: p.finish(scores=False)):.2f} ms")
shape = (nq, 5000)
bufs = (np.zeros(shape, np.uint32), np.zeros(shape, np.float32)); bufs[0][:] = 1; bufs[1][:] = 1
print(f"touched arrays, ids+scores: {t(lambda: p.finish(out=bufs)):.2f} ms")
print(f"touched arrays, ids only:   {t(lambda: p.finish(scores=False, out=bufs)):.2f} ms")
