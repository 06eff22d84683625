#!/usr/bin/env python3
"""Runs two drop-in builds of the reference CLI on the same synthetic workload and compares everything they write
(alignments.txt and every .SIFTprediction file).  usage: tools/compare_builds.py <build a> <build b> [n_queries] [n_targets] [cli args...]
e.g. tools/compare_builds.py sift4g sift4g_devsearch 200 50000 -t 32   (the reference's own CPU prefilter against the device's)"""
import filecmp, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sift4g_amd import synth
from sift4g_amd.seqio import write_fasta

a, b = sys.argv[1], sys.argv[2]
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 200
nt = int(sys.argv[4]) if len(sys.argv) > 4 else 50000
extra = sys.argv[5:]
tmp = tempfile.mkdtemp()
c = synth.make_config(2, nq, nt)
write_fasta(f"{tmp}/q.fa", [f"q{i}" for i in range(nq)], c["q_codes"], c["q_off"])
write_fasta(f"{tmp}/db.fa", [f"t{i}" for i in range(nt)], c["t_codes"], c["t_off"])
outs = []
for build in (a, b):
    out = f"{tmp}/{build}"
    os.makedirs(out)
    r = subprocess.run([f"{ROOT}/build/dropin/{build}", "-q", f"{tmp}/q.fa", "-d", f"{tmp}/db.fa", "--cards", "0", "--out", out, "--sub-results"] + extra,
                       cwd=out, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    outs.append(out)
names = sorted(set(os.listdir(outs[0])) | set(os.listdir(outs[1])))
match, mismatch, errors = filecmp.cmpfiles(outs[0], outs[1], names, shallow=False)
hits = sum(1 for ln in open(f"{outs[0]}/alignments.txt", "rb") if ln.strip() and not ln.startswith(b"#"))
print(f"{a} vs {b}: {nq} queries x {nt} targets, {len(names)} files, {len(match)} identical, {len(mismatch)} different, {len(errors)} missing on one side; "
      f"{hits} reported alignments in {a}'s alignments.txt")
for n in (mismatch + errors)[:10]:
    print("  differs:", n)
    if n in mismatch and n == "alignments.txt":        # which hits: lines only one side has
        la, lb = (set(open(f"{o}/{n}", "rb").read().split(b"\n")) for o in outs)
        for tag, only in ((a, la - lb), (b, lb - la)):
            for ln in sorted(only)[:8]:
                print(f"    only in {tag}: {ln.decode(errors='replace')}")
sys.exit(0 if not mismatch and not errors else 1)
