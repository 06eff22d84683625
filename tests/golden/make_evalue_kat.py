#!/usr/bin/env python3
"""Writes tests/golden/evalue_kat.json: known-answer vectors of the e-value path (createEValueParams + eValues,
/root/reference/sift4g/src/main.cpp:207-210, database_alignment.cpp:129-134) computed by the INDEPENDENT restatement
oracle/evalue_ref.py (written from Karlin-Altschul 1990 and Park-Sheetlin-Spouge 2012, not from the C++).  The C++ library
must reproduce every value to 1e-12 relative (tests/test_swsharp_host.py).  Both row conventions are recorded.

usage: python tests/golden/make_evalue_kat.py   (from the repository root; no GPU, no reference tree needed)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import evalue_ref as er   # noqa: E402

cases = []
scorers = [("BLOSUM_62", 10, 1), ("BLOSUM_62", 11, 1), ("BLOSUM_62", 11, 2), ("BLOSUM_45", 14, 2), ("BLOSUM_50", 13, 2),
           ("BLOSUM_80", 10, 1), ("BLOSUM_90", 10, 1), ("PAM_30", 9, 1), ("PAM_70", 10, 1), ("PAM_250", 14, 2)]
for matrix, go, ge in scorers:
    for conv in ("swsharp", "blast"):
        try:
            P = er.gumbel_parameters(matrix, go, ge, conv)
        except KeyError:
            continue                      # the restatement holds no row for this reading of the penalties
        for m, n in ((350, 340), (30, 3000), (5000, 45), (1, 1)):
            for db in (3091, 51_000_000_000):
                for score in (1, 25, 40, 64, 100, 150, 400, 12000):
                    cases.append([matrix, go, ge, conv, m, n, db, score, er.evalue(score, m, n, db, P)])
out = dict(columns=["matrix", "gap_open", "gap_extend", "convention", "m", "n", "db_residues", "score", "evalue"], note="e-values from oracle/evalue_ref.py (independent restatement of the published formulas); repr() of Python floats",
           generator="tests/golden/make_evalue_kat.py", cases=cases)
with open(os.path.join(HERE, "evalue_kat.json"), "w") as fh:
    json.dump(out, fh, separators=(",", ":"))
print(len(cases), "cases")
