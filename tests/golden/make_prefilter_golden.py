"""Generates tests/golden/prefilter.json: what the REFERENCE's own prefilter code returns on small seeded inputs.

Runs only where /root/reference exists: `make dropin-prefilter-dump` compiles the reference's database_search.cpp, hash.cpp and
utils.cpp where they lie behind tools/ref_prefilter_dump.cpp (its algorithm on this repo's Chain/FASTA substrate), this
script feeds it seeded FASTA files and stores, per case, the inputs' seed and parameters and the candidate ids it printed.
tests/test_oracle.py::test_prefilter_matches_the_reference_code rebuilds the same inputs from the seed and holds
oracle/kmer_prefilter.c to them; the GPU tests hold the device path to the oracle.

Cases are run at 1 and at 4 threads.  Where the cut (max_candidates) binds, the reference's choice among candidates whose
score equals the score at the cut differs between the two (and from the restatement's lowest-index rule), so a case stores the
ids common to both runs that lie strictly above the cut as `must` and the size of the list; where it does not bind the whole
list is stored and must match exactly.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LETTERS = "ARNDCQEGHILKMFPSTWYV"


def make_case(seed, nq, nt, repeats=False):
    """queries, targets as letter strings: random, with planted homologs (mutated copies, indels) and low-complexity repeats"""
    rng = np.random.default_rng(seed)

    def rand(n):
        return "".join(LETTERS[i] for i in rng.integers(0, 20, size=n))
    queries = [rand(int(n)) for n in rng.integers(30, 400, size=nq)]
    if nq > 2:
        queries[2] = rand(4)                                   # shorter than any k
    targets = [rand(int(n)) for n in rng.integers(3, 500, size=nt)]
    for q, s in enumerate(queries):
        for h in range(6):
            t = int(rng.integers(0, nt))
            m = list(s)
            for _ in range(int(len(m) * 0.08 * (h + 1) / 2)):
                m[int(rng.integers(0, len(m)))] = LETTERS[int(rng.integers(0, 20))]
            if h % 2 and len(m) > 20:
                cut = int(rng.integers(5, len(m) - 5))
                m = m[:cut] + list(rand(int(rng.integers(1, 9)))) + m[cut + int(rng.integers(0, 4)):]
            targets[t] = rand(int(rng.integers(0, 30))) + "".join(m) + rand(int(rng.integers(0, 30)))
    if repeats:
        queries[0] = "QA" * 40 + queries[0][:50] + "QQQQQQQQQQQQ" + "SPG" * 20
        targets[1] = "AQ" * 100
        targets[3] = "QQQQQQQQQQQQQQQQQQQQQQQQQ"
        targets[4] = "GSP" * 33 + "QA" * 10
        targets[6] = queries[0]
    return queries, targets


def write_fasta(path, prefix, seqs):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(f">{prefix}{i}\n")
            for o in range(0, len(s), 60):
                f.write(s[o:o + 60] + "\n")


def run_reference(queries, targets, k, max_candidates, threads):
    exe = os.path.join(ROOT, "build", "dropin", "ref_prefilter_dump")
    with tempfile.TemporaryDirectory() as d:
        write_fasta(os.path.join(d, "q.fa"), "q", queries)
        write_fasta(os.path.join(d, "db.fa"), "t", targets)
        r = subprocess.run([exe, os.path.join(d, "q.fa"), os.path.join(d, "db.fa"), str(k), str(max_candidates), str(threads)],
                           capture_output=True, text=True, check=True)
    lines = r.stdout.strip().split("\n")
    cells = int(lines[0].split()[1])
    lists = []
    for ln in lines[1:]:
        f = ln.split()
        assert int(f[1]) == len(f) - 2
        lists.append([int(x) for x in f[2:]])
    return cells, lists


CASES = [
    dict(name="k5_no_cut", seed=11, nq=6, nt=400, k=5, max_candidates=5000, repeats=False),
    dict(name="k5_repeats", seed=12, nq=5, nt=300, k=5, max_candidates=5000, repeats=True),
    dict(name="k4_no_cut", seed=13, nq=4, nt=300, k=4, max_candidates=5000, repeats=True),
    dict(name="k3_no_cut", seed=14, nq=4, nt=200, k=3, max_candidates=5000, repeats=True),
    dict(name="k3_cut_20", seed=15, nq=5, nt=400, k=3, max_candidates=20, repeats=False),
    dict(name="k5_cut_8", seed=16, nq=6, nt=2000, k=5, max_candidates=8, repeats=False),
    # larger databases: where nothing is cut off, only a SHA-256 of the lists is stored (`digest`)
    dict(name="k5_big_no_cut", seed=21, nq=30, nt=20000, k=5, max_candidates=5000, repeats=True, digest=True),
    dict(name="k4_big_no_cut", seed=22, nq=12, nt=8000, k=4, max_candidates=8000, repeats=True, digest=True),
    dict(name="k5_big_cut_100", seed=23, nq=24, nt=20000, k=5, max_candidates=100, repeats=False),
]


def lists_digest(lists):
    import hashlib
    return hashlib.sha256(json.dumps([[int(x) for x in l] for l in lists], separators=(",", ":")).encode()).hexdigest()


def main():
    import oracle
    subprocess.check_call(["make", "-s", "dropin-prefilter-dump"], cwd=ROOT)
    out = []
    for c in CASES:
        queries, targets = make_case(c["seed"], c["nq"], c["nt"], c["repeats"])
        cells1, l1 = run_reference(queries, targets, c["k"], c["max_candidates"], 1)
        cells4, l4 = run_reference(queries, targets, c["k"], c["max_candidates"], 4)
        assert cells1 == cells4 == sum(len(t) for t in targets)
        # the cut score of each query, from the restatement (only used to separate "strictly above" from "tied at the cut")
        pf = oracle.Prefilter([oracle.encode(q) for q in queries], c["k"], c["max_candidates"])
        pf.add([oracle.encode(t) for t in targets])
        ids, counts, scores, cut = pf.finish()
        entry = dict(c, cells=cells1, lists=[])
        if c.get("digest"):
            assert l1 == l4 and all(x == 0.0 for x in cut), c["name"]
            entry["lists_sha256"] = lists_digest(l1)
            entry["list_sizes"] = [len(x) for x in l1]
            out.append(entry)
            print(c["name"], "cells", cells1, "list sizes", entry["list_sizes"][:8], "...")
            continue
        for q in range(c["nq"]):
            if cut[q] == 0.0:                              # nothing cut off: one answer
                assert l1[q] == l4[q], (c["name"], q)
                entry["lists"].append(dict(exact=l1[q]))
            else:
                above = {int(i) for i, s in zip(ids[q, :counts[q]], scores[q, :counts[q]]) if s > cut[q]}
                assert above <= set(l1[q]) and above <= set(l4[q]), (c["name"], q)
                entry["lists"].append(dict(size=len(l1[q]), must=sorted(set(l1[q]) & set(l4[q]) & above),
                                           differs_with_threads=l1[q] != l4[q]))
                assert len(l1[q]) == len(l4[q])
        out.append(entry)
        print(c["name"], "cells", cells1, "list sizes", [len(x) for x in l1])
    with open(os.path.join(ROOT, "tests", "golden", "prefilter.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
