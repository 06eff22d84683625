#!/usr/bin/env python3
"""SHA-256 of a substitution matrix in a canonical text form, for comparing this repo's typed-in NCBI tables
(sift4g_amd/csrc/matrices.h) with a matrix file from NCBI's distribution (ftp.ncbi.nih.gov/blast/matrices) -- none is present
in this image, so the comparison is left to whoever has one.

Canonical form: symbol order ARNDCQEGHILKMFPSTWYVBZX*, 24 rows of 24 decimal integers separated by single spaces, rows ended
by "\n".  A file in NCBI's format (comment lines starting with '#', a header line of column letters, rows starting with their
letter, extra columns such as J allowed) is reduced to that form before hashing.

  tools/matrix_digest.py BLOSUM62.txt            digest of a matrix file
  tools/matrix_digest.py --builtin               digests of every built-in table (what tests/golden/matrix_sha256.json holds)"""
import hashlib
import json
import os
import sys

ORDER = "ARNDCQEGHILKMFPSTWYVBZX*"


def canonical_from_table26(table26):
    """26 x 26 letter-indexed table (code = letter - 'A'; '*' is not representable and is left out: 23 x 23 symbols)"""
    syms = ORDER[:-1]
    return "".join(" ".join(str(int(table26[ord(a) - 65][ord(b) - 65])) for b in syms) + "\n" for a in syms)


def canonical_from_ncbi_file(text):
    rows, cols = {}, None
    for line in text.splitlines():
        if not line.strip() or line.lstrip().startswith("#"):
            continue
        parts = line.split()
        if cols is None:
            cols = parts
            continue
        rows[parts[0]] = dict(zip(cols, (int(x) for x in parts[1:])))
    syms = ORDER[:-1]
    return "".join(" ".join(str(rows[a][b]) for b in syms) + "\n" for a in syms)


def digest(canon):
    return hashlib.sha256(canon.encode()).hexdigest()


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "--builtin":
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from sift4g_amd.device import matrix_by_name
        names = ["BLOSUM_45", "BLOSUM_50", "BLOSUM_62", "BLOSUM_80", "BLOSUM_90", "PAM_30", "PAM_70", "PAM_250"]
        print(json.dumps({n: digest(canonical_from_table26(matrix_by_name(n))) for n in names}, indent=1))
    elif len(sys.argv) == 2:
        print(digest(canonical_from_ncbi_file(open(sys.argv[1]).read())))
    else:
        sys.exit(__doc__)
