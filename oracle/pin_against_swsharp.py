#!/usr/bin/env python3
"""oracle/pin_against_swsharp.py -- TEST INFRASTRUCTURE: the one command that would pin this repo's oracle to the reference's real scorer.

    make -C oracle _ref SWSHARP_SRC=<checkout of mkorpar/swsharp>        # builds oracle/_ref/{pin_dump_ref, sift4g_ref} (CPU build)
    make pin-dump                                                          # builds build/pin_dump_ours (this repo's libswsharp)
    python oracle/pin_against_swsharp.py [--ours gpu|oracle] [--out report.json]

What it compares (the reference's call of database_alignment.cpp:83-86, through oracle/pin_dump.cpp compiled against each library):

  1. scores        int32 local alignment scores of every (query, target) pair                     -- must be equal
  2. e-values      createEValueParams / eValues (main.cpp:210; database_alignment.cpp:129-134)     -- 1e-9 relative
  3. order         the order alignDatabase returns the survivors in                                -- must be equal
  4. tracebacks    query start, target start, move string (select_alignments.cpp:244-300)          -- must be equal
  5. the CLI       oracle/_ref/sift4g_ref (the reference's unmodified sources on upstream swsharp) on test_files/ and on the synthetic set
                   of tests/dropin_synth.py: alignments.txt and every .SIFTprediction == tests/golden/dropin_* (the bytes this repo's
                   GPU path and its oracle stand-in both write)

on four inputs: the reference's fixture, the synthetic set, a set that uses all 26 letter codes (B, J, O, U, X, Z: the J/O/U policy of
INTEGRATION.md section 5), and a low-complexity set that is full of traceback ties; under BLOSUM_62 10/1 and 11/1, BLOSUM_45 14/2,
BLOSUM_80 10/1.  Where a column differs it re-runs THIS repo's side under the documented switches (S4G_EVALUE_CONVENTION,
S4G_TRACEBACK_TIES, S4G_JOU_POLICY, S4G_SELECTION_TIES) and reports which setting, if any, removes the difference -- e.g. whether
upstream looks the Gumbel row of a 10/1 scorer up as (10, 1) (this repo's default, "as far as upstream is recalled") or as BLAST's
(9, 1) (what the Monte-Carlo of tests/test_gumbel_mc.py says the scorer's statistics are).

`--ours oracle` (the default without a GPU) runs this repo's side on tests/cpu_shim -- libswsharp's host code over oracle/sw_oracle.c --
so the ORACLE itself is what gets pinned; `--ours gpu` runs the HIP path.  Exit code: 0 everything equal, 1 differences (listed),
77 oracle/_ref is not built.  In this image that is always 77: upstream's sources are an absent submodule (/root/reference/.gitmodules:1-3,
vendor/swsharp empty) and there is no network; parity stays UNPINNED until somebody runs this where a checkout exists.
"""
import argparse
import itertools
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "pin_dump_ref")
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "sift4g_ref")
OURS_DUMP = os.path.join(ROOT, "build", "pin_dump_ours")
SHIM_DIR = os.path.join(ROOT, "tests", "cpu_shim")
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCORERS = (("BLOSUM_62", 10, 1), ("BLOSUM_62", 11, 1), ("BLOSUM_45", 14, 2), ("BLOSUM_80", 10, 1))
SWITCHES = {"S4G_EVALUE_CONVENTION": ("swsharp", "blast"), "S4G_TRACEBACK_TIES": ("up", "left"),
            "S4G_JOU_POLICY": ("x", "similar", "stop"), "S4G_SELECTION_TIES": ("index", "reverse", "name")}


def ours_env(mode, switches=None):
    env = dict(os.environ)
    for k in SWITCHES:
        env.pop(k, None)
    if mode == "oracle":
        subprocess.run(["make", "-s", "-C", SHIM_DIR], check=True)
        env["LD_LIBRARY_PATH"] = os.path.join(SHIM_DIR, "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    for k, v in (switches or {}).items():
        if v != SWITCHES[k][0]:                       # the first value of every switch is the default: leave the variable unset
            env[k] = v
    return env


def run_dump(exe, query, database, scorer, env=None, max_alignments=400, max_evalue=1e9):
    r = subprocess.run([exe, query, database, scorer[0], str(scorer[1]), str(scorer[2]), str(max_alignments), repr(max_evalue)],
                       capture_output=True, text=True, env=env, timeout=3600)
    if r.returncode != 0:
        raise RuntimeError(f"{exe} exited with {r.returncode}: {r.stderr[-1500:]}")
    return parse_dump(r.stdout)


def parse_dump(text):
    """-> list of records (query, target, score, evalue, q_start, t_start, path) in output order"""
    out = []
    for line in text.splitlines():
        if not line or line.startswith("#"):
            continue
        f = line.split("\t")
        out.append((f[0], f[1], int(f[2]), float(f[3]), int(f[4]), int(f[5]), f[6] if len(f) > 6 else ""))
    return out


def compare_dumps(ref, ours, rel=1e-9):
    """column-by-column differences of two dumps -> dict(scores, evalues, order, tracebacks, missing), each a list of examples"""
    a = {(r[0], r[1]): r for r in ref}
    b = {(r[0], r[1]): r for r in ours}
    d = {"missing": sorted(set(a) ^ set(b))[:10], "scores": [], "evalues": [], "tracebacks": [], "order": []}
    for k in sorted(set(a) & set(b)):
        x, y = a[k], b[k]
        if x[2] != y[2]:
            d["scores"].append((k, x[2], y[2]))
            continue                                   # a different score explains the other columns
        if abs(x[3] - y[3]) > rel * max(abs(x[3]), abs(y[3]), 1e-300):
            d["evalues"].append((k, x[3], y[3]))
        if (x[4], x[5], x[6]) != (y[4], y[5], y[6]):
            d["tracebacks"].append((k, x[4:7], y[4:7]))
    ra, rb = [(r[0], r[1]) for r in ref if (r[0], r[1]) in b], [(r[0], r[1]) for r in ours if (r[0], r[1]) in a]
    d["order"] = [(i, p, q) for i, (p, q) in enumerate(zip(ra, rb)) if p != q][:10]
    d["counts"] = {k: len(v) for k, v in d.items() if isinstance(v, list)}
    for k in ("scores", "evalues", "tracebacks"):
        d[k] = d[k][:10]
    d["equal"] = not any(d["counts"].values())
    return d


def explain(ref, query, database, scorer, mode, base):
    """re-run this repo's side under the documented switches until the differing columns agree -> (switches, comparison) or (None, base)"""
    need = [k for k, col in (("S4G_EVALUE_CONVENTION", "evalues"), ("S4G_TRACEBACK_TIES", "tracebacks"), ("S4G_JOU_POLICY", "scores"),
                             ("S4G_SELECTION_TIES", "order")) if base["counts"].get(col)]
    for combo in itertools.product(*[SWITCHES[k] for k in need]):
        sw = dict(zip(need, combo))
        if all(v == SWITCHES[k][0] for k, v in sw.items()):
            continue
        cmp_ = compare_dumps(ref, run_dump(OURS_DUMP, query, database, scorer, ours_env(mode, sw)))
        if cmp_["equal"]:
            return sw, cmp_
    return None, base


def make_inputs(workdir):
    """the four input pairs (query FASTA, database FASTA); the synthetic ones come from the repo's seeded generators"""
    sys.path.insert(0, ROOT)
    import numpy as np
    from sift4g_amd import synth
    from sift4g_amd.seqio import write_fasta
    tf = os.path.join(GOLDEN, "test_files")
    sets = {"fixture": (os.path.join(tf, "query.fasta"), os.path.join(tf, "sample_protein_database.fa"))}
    c = synth.make_config(2, 8, 1500, homologs_per_query=30)                   # tests/dropin_synth.py's set
    q, d = os.path.join(workdir, "synth_q.fa"), os.path.join(workdir, "synth_db.fa")
    write_fasta(q, [f"q{i}" for i in range(8)], c["q_codes"], c["q_off"]); write_fasta(d, [f"t{i}" for i in range(1500)], c["t_codes"], c["t_off"])
    sets["synthetic"] = (q, d)
    rng = np.random.default_rng(20260406)

    def flat(seqs):
        off = np.zeros(len(seqs) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(s) for s in seqs]).astype(np.uint64)
        return np.concatenate(seqs).astype(np.uint8), off
    qs = [rng.integers(0, 26, size=int(n)).astype(np.uint8) for n in rng.integers(40, 300, size=6)]
    ts = [rng.integers(0, 26, size=int(n)).astype(np.uint8) for n in rng.integers(20, 400, size=300)]
    for k in range(0, 300, 5):                                                 # homologs, so that paths and e-values are non-trivial
        m = qs[(k // 5) % 6].copy(); idx = rng.choice(len(m), size=len(m) // 5, replace=False); m[idx] = rng.integers(0, 26, size=len(idx)); ts[k] = m
    q, d = os.path.join(workdir, "letters_q.fa"), os.path.join(workdir, "letters_db.fa")
    write_fasta(q, [f"q{i}" for i in range(len(qs))], *flat(qs)); write_fasta(d, [f"t{i}" for i in range(len(ts))], *flat(ts))
    sets["all 26 letters"] = (q, d)
    qs = [np.tile(rng.integers(0, 4, size=3).astype(np.uint8) * 5, int(n)) for n in rng.integers(15, 60, size=5)]      # repeats of a 3-mer: ties everywhere
    ts = [np.tile(rng.integers(0, 4, size=int(p)).astype(np.uint8) * 5, int(n)) for p, n in zip(rng.integers(2, 5, size=120), rng.integers(10, 80, size=120))]
    q, d = os.path.join(workdir, "ties_q.fa"), os.path.join(workdir, "ties_db.fa")
    write_fasta(q, [f"q{i}" for i in range(len(qs))], *flat(qs)); write_fasta(d, [f"t{i}" for i in range(len(ts))], *flat(ts))
    sets["low complexity (ties)"] = (q, d)
    return sets


def cli_leg(workdir):
    """the reference's CLI on upstream swsharp against the committed outputs of this repo's path"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    out = {}
    tf = os.path.join(GOLDEN, "test_files")
    d = os.path.join(workdir, "cli_fixture"); os.makedirs(d, exist_ok=True)
    r = subprocess.run([REF_CLI, "-q", os.path.join(tf, "query.fasta"), "--subst", tf, "-d", os.path.join(tf, "sample_protein_database.fa"),
                        "--out", d, "--sub-results"], capture_output=True, text=True, cwd=d, timeout=3600)
    out["fixture"] = {"rc": r.returncode, "files": {}}
    for n in ("alignments.txt", "LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction"):
        p = os.path.join(d, n)
        out["fixture"]["files"][n] = os.path.exists(p) and open(p, "rb").read() == open(os.path.join(GOLDEN, "dropin_predictions", n), "rb").read()
    import dropin_synth
    d = os.path.join(workdir, "cli_synth"); os.makedirs(d, exist_ok=True)
    try:
        got = dropin_synth.run(REF_CLI, d)
        out["synthetic"] = {"rc": 0, "files": {"alignments.txt": got == open(os.path.join(GOLDEN, "dropin_synth", "alignments.txt"), "rb").read()}}
    except AssertionError as exc:
        out["synthetic"] = {"rc": 1, "error": str(exc)[-500:], "files": {"alignments.txt": False}}
    out["equal"] = all(all(v["files"].values()) and v["rc"] == 0 for v in out.values() if isinstance(v, dict))
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--ours", choices=["auto", "gpu", "oracle"], default="auto")
    ap.add_argument("--out", default=os.path.join(ROOT, "oracle", "_ref", "pin_report.json"))
    args = ap.parse_args()
    if not (os.path.exists(REF_DUMP) and os.path.exists(REF_CLI)):
        print("pin_against_swsharp: oracle/_ref/ is not built -- upstream swsharp is an absent submodule here (/root/reference/.gitmodules:1-3, "
              "vendor/swsharp empty, no network).  PARITY UNPINNED.  With a checkout: make -C oracle _ref SWSHARP_SRC=<dir> && make pin-dump && "
              "python oracle/pin_against_swsharp.py", file=sys.stderr)
        return 77
    if not os.path.exists(OURS_DUMP):
        subprocess.run(["make", "-C", ROOT, "pin-dump"], check=True)
    mode = args.ours
    if mode == "auto":
        try:
            import torch
            mode = "gpu" if torch.cuda.device_count() > 0 else "oracle"
        except ImportError:
            mode = "oracle"
    report = {"ours": mode, "dumps": [], "switches_that_match_upstream": {}}
    ok = True
    with tempfile.TemporaryDirectory() as wd:
        for name, (q, d) in make_inputs(wd).items():
            for scorer in SCORERS:
                ref = run_dump(REF_DUMP, q, d, scorer)
                cmp_ = compare_dumps(ref, run_dump(OURS_DUMP, q, d, scorer, ours_env(mode)))
                entry = {"input": name, "scorer": scorer, "records": len(ref), "default": cmp_}
                if not cmp_["equal"]:
                    sw, after = explain(ref, q, d, scorer, mode, cmp_)
                    entry["equal_under"] = sw
                    if sw:
                        for k, v in sw.items():
                            report["switches_that_match_upstream"].setdefault(k, set()).add(v)
                    ok = False
                report["dumps"].append(entry)
                print(f"{name:24s} {scorer[0]} {scorer[1]}/{scorer[2]}: {len(ref)} records, " +
                      ("EQUAL" if cmp_["equal"] else f"DIFFERENT {cmp_['counts']}" + (f" -> equal under {entry['equal_under']}" if entry.get("equal_under") else " -> no documented switch explains it")))
        report["cli"] = cli_leg(wd)
        ok = ok and report["cli"]["equal"]
        print("reference CLI on upstream swsharp vs tests/golden/dropin_*: " + ("EQUAL" if report["cli"]["equal"] else f"DIFFERENT {report['cli']}"))
    report["switches_that_match_upstream"] = {k: sorted(v) for k, v in report["switches_that_match_upstream"].items()}
    report["pinned"] = ok
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(report, open(args.out, "w"), indent=1, default=str)
    print(("PINNED: every column equal to upstream swsharp with the default settings" if ok else
           f"NOT EQUAL with the default settings; switches that match upstream: {report['switches_that_match_upstream']}") + f"  (report: {args.out})")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
