"""Third, structurally different Smith-Waterman scorer -- TEST INFRASTRUCTURE ONLY (importers: tests/).

oracle/sw_oracle.c walks the DP row by row with -infinity-initialised E/F; oracle/sw_simd.c walks it column by column,
16 targets at a time, in saturating int16; the HIP kernels walk strips of 32 rows with E/F clamped at 0, or sweep
anti-diagonals over 64 lanes.  This one sweeps ANTI-DIAGONALS of one pair with whole-array numpy operations in int64:
the recurrence of SURVEY.md 8a row a-R (the definition, not a reference file; gap of k costs open + (k-1) extend,
/root/reference/sift4g/src/main.cpp:297-303) written a third time, so that an error of reading in any one of the
forms would show up as a disagreement (tests/test_oracle.py triangulates the three on random and adversarial pairs).
"""
import numpy as np

NEG = -(1 << 40)


def sw_score(q, t, M, gap_open=10, gap_extend=1):
    """optimal local alignment score of code arrays q, t under the 26x26 table M"""
    q = np.asarray(q, dtype=np.int64)
    t = np.asarray(t, dtype=np.int64)
    lq, lt = len(q), len(t)
    if lq == 0 or lt == 0:
        return 0
    M = np.asarray(M, dtype=np.int64)
    S = M[q][:, t]                              # S[i, j] = M[q_i][t_j]
    # cell (i, j), 0-based, lies on anti-diagonal d = i + j; arrays are indexed by i (row), length lq
    H1 = np.zeros(lq, dtype=np.int64)           # H on diagonal d-1
    H2 = np.zeros(lq, dtype=np.int64)           # H on diagonal d-2
    E1 = np.full(lq, NEG, dtype=np.int64)       # E on diagonal d-1 (same row, previous column)
    F1 = np.full(lq, NEG, dtype=np.int64)       # F on diagonal d-1 (previous row, same column)
    valid1 = np.zeros(lq, dtype=bool)           # which rows had a cell on diagonal d-1 / d-2
    valid2 = np.zeros(lq, dtype=bool)
    rows = np.arange(lq)
    best = 0
    for d in range(lq + lt - 1):
        j = d - rows
        on = (j >= 0) & (j < lt)
        # left neighbour (i, j-1): diagonal d-1, same row
        left_ok = on & valid1
        E = np.where(left_ok, np.maximum(E1 - gap_extend, H1 - gap_open), NEG)
        # upper neighbour (i-1, j): diagonal d-1, row i-1
        upH = np.concatenate(([0], H1[:-1]))
        upF = np.concatenate(([NEG], F1[:-1]))
        up_ok = on & np.concatenate(([False], valid1[:-1]))
        F = np.where(up_ok, np.maximum(upF - gap_extend, upH - gap_open), NEG)
        # diagonal neighbour (i-1, j-1): diagonal d-2, row i-1; outside the matrix H = 0
        dgH = np.concatenate(([0], H2[:-1]))
        dg_ok = np.concatenate(([False], valid2[:-1]))
        diag = np.where(dg_ok, dgH, 0)
        sc = np.where(on, S[rows, np.clip(j, 0, lt - 1)], 0)
        H = np.where(on, np.maximum(np.maximum(0, diag + sc), np.maximum(E, F)), 0)
        if on.any():
            best = max(best, int(H[on].max()))
        H2, valid2 = H1, valid1
        H1, E1, F1, valid1 = H, E, F, on
    return best


def scan(queries, targets, M, gap_open=10, gap_extend=1):
    return np.array([[sw_score(q, t, M, gap_open, gap_extend) for t in targets] for q in queries], dtype=np.int32)
