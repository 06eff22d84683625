#!/usr/bin/env python3
"""s4g_scan for FEW queries against a large database: the library's own routing against the forced forms (EXPERIMENTS.md R6.2).

  python tools/few_query_scan.py <n_targets> <query length>x<count> [<length>x<count> ...]
  e.g.  python tools/few_query_scan.py 100000 5000x1 5000x3 5000x8 20000x1 350x4        (long queries: speculative pipeline, twelve waves)
        python tools/few_query_scan.py 570000 350x1 350x2 350x3 1000x1 350x5            (a lone query pairs groups)

Per shape: ms per call (best of three) and GCUPS for   auto        the library's choice
                                                      pipe4/12    the block pipeline with four / twelve waves per block (S4G_SCAN_PIPE_WAVES)
                                                      norepeat... pipe4 with a lone query repeated in both halves (S4G_SCAN_PIPE_NO_GROUP_PAIRS)
                                                      scan        the scan kernels alone (no_pairs)
and whether all forms wrote the same matrix.  Database: the bench generator (synth.make_config(3, ...): lognormal lengths clipped at 3000)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import sift4g_amd as s4g
from sift4g_amd import device as s4gd, synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
shapes = [tuple(int(x) for x in a.split("x")) for a in (sys.argv[2:] or ["350x1", "350x4", "5000x1", "5000x3"])]
rng = np.random.default_rng(78)
std = np.array([0, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 15, 16, 17, 18, 19, 21, 22, 24], dtype=np.uint8)
scorer = s4g.Scorer("BLOSUM_62", 10, 1)
cfg = synth.make_config(3, n_queries=2, n_targets=nt)
db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])
forms = (("auto", dict(), {}), ("pipe4", dict(block_pipe=True), {"S4G_SCAN_PIPE_WAVES": "4"}), ("pipe12", dict(block_pipe=True), {"S4G_SCAN_PIPE_WAVES": "12"}),
         ("pipe4 repeated query", dict(block_pipe=True), {"S4G_SCAN_PIPE_WAVES": "4", "S4G_SCAN_PIPE_NO_GROUP_PAIRS": "1"}), ("scan", dict(no_pairs=True), {}))
for length, count in shapes:
    q = s4g.QuerySet([std[rng.integers(0, 20, size=length)] for _ in range(count)])
    out = torch.empty((q.n, db.n), dtype=torch.int32, device="cuda:0")
    cells = s4gd.scan_cells(db, q)
    res, outs = [], []
    for name, mode, env in forms:
        for k in ("S4G_SCAN_PIPE_WAVES", "S4G_SCAN_PIPE_NO_GROUP_PAIRS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s4g.scan(db, q, scorer, out, **mode); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        outs.append(out.clone())
        st = s4gd.scan_stats(0)
        res.append(f"{name} {best * 1e3:.2f} ms {cells / best / 1e9:.0f} GCUPS" + (f" [{', '.join(k for k, v in st.items() if v)}]" if name == "auto" else ""))
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{count} x {length} residues x {nt} targets ({cells:.2e} cells): " + " | ".join(res) + (" | all equal" if same else " | DIFFERENT RESULTS"))
