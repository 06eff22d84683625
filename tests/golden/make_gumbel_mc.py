#!/usr/bin/env python3
"""Monte-Carlo pin of the typed Gumbel rows of sift4g_amd/csrc/swsharp/evalue.cpp.

The gapped rows of createEValueParams (lambda, K, alpha ...) are simulation results published with NCBI BLAST and were typed in
from memory: nothing in this image holds a copy.  What the image does hold is a scorer that visits 10^13 cells per second, so
the rows are checked the way they were made: by scoring random sequence pairs.

For every typed row (matrix, existence g, extension e) -- i.e. for the scorer whose gap of k residues costs g + k e, which in this
library's (and swsharp's) parameters is open = g + e, extend = e -- and for each sequence length L in --lengths:

  * Q random queries and T random targets of L residues, residues i.i.d. from the Robinson & Robinson background the published
    tables were computed with (torch generator on the device, seeded);
  * the optimal local scores of all Q x T pairs by s4g_scan (the product path, every tier exact; `--check` re-scores a sample
    of pairs with the CPU oracle);
  * the histogram of the scores, kept per (query group, target group) block of an 8 x 8 cut of the grid: blocks on one
    "diagonal" (i, (i + s) mod 8) share no sequence, so the spread of per-block fits over a diagonal is an honest error
    estimate although pairs inside a block share sequences.

The fit (tests/gumbel_fit.py, shared with tests/test_gumbel_mc.py): with F(y) = P(S < y) = exp(-E(y)) and Spouge's finite-size
corrected E(y) = area(L, L, y) K exp(-lambda y) -- the same area formula evalue.cpp evaluates, with the ROW'S OWN alpha / sigma /
a -- lambda and K are found by maximum likelihood on the tail bins y >= y0 (y0: the empirical P(S >= y0) <= 0.02).

Output: tests/golden/gumbel_mc.json -- per row the tail histogram (so the fit can be redone on the CPU), the fitted (lambda, K)
with block-spread standard errors, and the typed values as they stood when the file was made.

usage (GPU box):  python3 tests/golden/make_gumbel_mc.py [--lengths 350 1000] [--pairs-side 8192 4096] [--matrices BLOSUM_62 ...] [--out path]
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gumbel_fit as gf          # noqa: E402

G = 8                            # the Q x T grid is cut into G x G blocks
AA20 = "ARNDCQEGHILKMFPSTWYV"
ROBINSON = [0.07805, 0.05129, 0.04487, 0.05364, 0.01925, 0.04264, 0.06295, 0.07377, 0.02199, 0.05142,
            0.09019, 0.05744, 0.02243, 0.03856, 0.05203, 0.07120, 0.05841, 0.01330, 0.03216, 0.06441]


def random_codes(n_seqs, length, seed, dev):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    cum = torch.tensor(np.cumsum(ROBINSON)[:-1], dtype=torch.float32, device=dev)
    letters = torch.tensor([ord(c) - 65 for c in AA20], dtype=torch.uint8, device=dev)
    u = torch.rand(n_seqs * length, device=dev, generator=g)
    codes = letters[torch.bucketize(u, cum, right=True).clamp_(0, 19)]
    offsets = torch.arange(0, (n_seqs + 1) * length, length, dtype=torch.int64, device=dev)
    return codes.contiguous(), offsets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lengths", type=int, nargs="+", default=[350, 1000])
    ap.add_argument("--pairs-side", type=int, nargs="+", default=[8192, 4096], help="Q = T per length (a multiple of 8 x 64)")
    ap.add_argument("--matrices", nargs="+", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "gumbel_mc.json"))
    ap.add_argument("--check", type=int, default=300, help="pairs per row and length re-scored by the CPU oracle")
    ap.add_argument("--seed", type=int, default=0x51F74A00)
    args = ap.parse_args()
    assert len(args.lengths) == len(args.pairs_side)

    import torch
    import sift4g_amd as s4g
    from sift4g_amd import device as s4gd
    import oracle
    dev = torch.device("cuda", 0)
    tables = gf.typed_tables(os.path.join(ROOT, "sift4g_amd", "csrc", "swsharp", "evalue.cpp"))
    rows_out = []
    t_all = time.perf_counter()
    cells_all = 0.0
    sets = {}
    for L, side in zip(args.lengths, args.pairs_side):            # the same sequences serve every row (built once per length)
        qc, qo = random_codes(side, L, args.seed + L, dev)
        tc, to = random_codes(side, L, args.seed + 7919 * L, dev)
        db = s4g.Database.from_device(tc, to)
        qh, qoh = qc.cpu().numpy(), qo.cpu().numpy().astype(np.uint64)
        q = s4g.QuerySet(codes=qh, offsets=qoh)
        sets[L] = dict(db=db, q=q, q_codes=qh, q_off=qoh, t_codes=tc.cpu().numpy(), t_off=to.cpu().numpy().astype(np.uint64),
                       scores=torch.empty((side, side), dtype=torch.int32, device=dev), side=side)
    for matrix, rows in tables.items():
        if args.matrices and matrix not in args.matrices:
            continue
        M = s4gd.matrix_by_name(matrix)
        ungapped = rows[0]
        for row in rows[1:]:
            g_, e_ = row["open"], row["extension"]
            scorer = s4g.Scorer(matrix, g_ + e_, e_)              # gap of k costs (g + e) + (k - 1) e = g + k e
            entry = dict(matrix=matrix, existence=g_, extension=e_, scorer_open=g_ + e_, scorer_extend=e_,
                         typed={k: row[k] for k in ("lambda", "K", "H", "a", "Alpha", "Sigma")},
                         typed_ungapped={k: ungapped[k] for k in ("a", "Alpha")}, lengths={})
            for L in args.lengths:
                S = sets[L]
                side = S["side"]
                t0 = time.perf_counter()
                s4g.scan(S["db"], S["q"], scorer, S["scores"])
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                cells_all += float(side) * side * L * L
                sc = S["scores"]
                smax = int(sc.max()) + 1
                blk = side // G
                bi = (torch.arange(side, device=dev) // blk)
                block_id = (bi[:, None] * G + bi[None, :]).to(torch.int64)
                hist = torch.bincount((block_id * smax + sc.to(torch.int64)).flatten(), minlength=G * G * smax)
                hist = hist.view(G, G, smax).cpu().numpy()
                bad = None
                if args.check:
                    rng = np.random.default_rng(L + g_ * 100 + e_)
                    pq = rng.integers(0, side, args.check).astype(np.uint32)
                    pt = rng.integers(0, side, args.check).astype(np.uint32)
                    want = oracle.pairs_flat(S["q_codes"], S["q_off"], S["t_codes"], S["t_off"], pq, pt, M.astype(np.int32), g_ + e_, e_,
                                             os.cpu_count() or 8)
                    have = sc[torch.from_numpy(pq.astype(np.int64)).to(dev), torch.from_numpy(pt.astype(np.int64)).to(dev)].cpu().numpy()
                    bad = int((want != have).sum())
                    assert bad == 0, (matrix, g_, e_, L, bad)
                fit = gf.fit_row(hist, L, row, ungapped)
                # keep the tail of the all-pairs histogram (from the first score whose tail probability is below 0.2: enough to redo
                # the fit anywhere) and the per-block fits (for the error estimate)
                total = hist.sum(axis=(0, 1))
                tail = np.cumsum(total[::-1])[::-1] / total.sum()
                y_keep = int(np.argmax(tail <= 0.2))
                entry["lengths"][str(L)] = dict(
                    pairs=int(total.sum()), side=side, blocks=G, y_first=y_keep, below=int(total[:y_keep].sum()),
                    hist=total[y_keep:].tolist(), scan_s=round(dt, 3), oracle_checked=args.check, oracle_mismatches=bad, fit=fit)
                print(f"{matrix} {g_}/{e_} L={L}: lambda {fit['lambda']:.4f} +- {fit['lambda_se']:.4f} (typed {row['lambda']}), "
                      f"K {fit['K']:.4f} x/ {np.exp(fit['lnK_se']):.3f} (typed {row['K']}), tail ratio typed/observed "
                      f"{fit['tail_ratio_typed']:.3f} at y={fit['y_ref']}; scan {dt:.2f} s", flush=True)
            rows_out.append(entry)
    meta = dict(made_by="tests/golden/make_gumbel_mc.py", seed=args.seed, lengths=args.lengths, pairs_side=args.pairs_side, background="Robinson & Robinson 1991",
                scorer="s4g_scan (libs4g_hip.so) on " + torch.cuda.get_device_name(0), wall_s=round(time.perf_counter() - t_all, 1),
                cells=cells_all, model="P(S < y) = exp(-area(L, L, y) K exp(-lambda y)), area = Spouge finite-size correction with the row's "
                "own a / alpha / sigma (evalue.cpp); maximum likelihood on the bins y >= y0, P(S >= y0) <= 0.02")
    json.dump(dict(meta=meta, rows=rows_out), open(args.out, "w"), separators=(",", ":"))
    print(f"wrote {args.out}: {len(rows_out)} rows, {cells_all:.3e} cells in {meta['wall_s']} s")


if __name__ == "__main__":
    main()
