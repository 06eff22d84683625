"""Regenerates tests/golden/cfg1_scores.json: the 2 x 9 score matrix of BASELINE.json configs[0]
(test_files/query.fasta x sample_protein_database.fa, BLOSUM_62, gap 10/1) from the CPU ORACLE.

These are oracle outputs, not reference outputs: the reference ships no expected results for this path
and its scorer (swsharp) is an absent submodule ("parity unpinned", see oracle/sw_oracle.c).
The FASTA/.subst files under tests/golden/test_files/ are the reference's own input fixtures (data).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from sift4g_amd.seqio import read_fasta  # noqa: E402


def main():
    M = oracle.blosum62()
    queries = read_fasta(os.path.join(HERE, "test_files", "query.fasta"))
    targets = read_fasta(os.path.join(HERE, "test_files", "sample_protein_database.fa"))
    scores = oracle.scan([c for _, c in queries], [c for _, c in targets], M, 10, 1)
    out = dict(
        comment="oracle-generated (parity unpinned); BLOSUM_62 gap_open 10 gap_extend 1; rows = queries, cols = database entries",
        queries=[n for n, _ in queries], query_lengths=[len(c) for _, c in queries],
        targets=[n for n, _ in targets], target_lengths=[len(c) for _, c in targets],
        scores=scores.tolist())
    with open(os.path.join(HERE, "cfg1_scores.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["scores"]))


if __name__ == "__main__":
    main()
