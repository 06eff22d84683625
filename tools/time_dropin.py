#!/usr/bin/env python3
"""Runs the drop-in reference CLI (build/dropin/sift4g) on a synthetic workload and reports wall time per pipeline stage
(stage boundaries = the reference's own stderr banners).  usage: tools/time_dropin.py [n_queries] [n_targets] [extra cli args...]
S4G_DROPIN_BIN=sift4g_batched selects the build with the batched alignment wrapper (make dropin-batched); S4G_DROPIN_DATA=<dir> keeps the
generated FASTA files there and reuses them (several builds timed on one input without paying the generator each time)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sift4g_amd import synth
from sift4g_amd.seqio import write_fasta

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
extra = sys.argv[3:]
tmp = tempfile.mkdtemp()
data = os.environ.get("S4G_DROPIN_DATA") or tmp
os.makedirs(data, exist_ok=True)
if not (os.path.exists(f"{data}/q_{nq}.fa") and os.path.exists(f"{data}/db_{nt}.fa")):
    c = synth.make_config(2, nq, nt)
    write_fasta(f"{data}/q_{nq}.fa", [f"q{i}" for i in range(nq)], c["q_codes"], c["q_off"])
    write_fasta(f"{data}/db_{nt}.fa", [f"t{i}" for i in range(nt)], c["t_codes"], c["t_off"])
os.makedirs(f"{tmp}/out")
p = subprocess.Popen([f"{ROOT}/build/dropin/{os.environ.get('S4G_DROPIN_BIN', 'sift4g')}", "-q", f"{data}/q_{nq}.fa", "-d", f"{data}/db_{nt}.fa", "--cards", "0", "--out", f"{tmp}/out"] + extra,
                     stderr=subprocess.PIPE, cwd=f"{tmp}/out")
t0 = time.time(); last = t0; stage = "start"; buf = b""; tail = []
while True:
    ch = p.stderr.read(1)
    if not ch:
        break
    if ch in b"\r\n":
        line = buf.decode(errors="ignore").strip(); buf = b""
        if line.startswith("[s4g profile]"):
            print(line)
        elif line and not line.startswith("*"):
            tail = (tail + [line])[-8:]
        if line.startswith("** ") and line.endswith("**"):
            now = time.time(); print(f"{now - last:8.2f} s  {stage}"); last = now; stage = line.strip("* ")
    else:
        buf += ch
p.wait()
if p.returncode != 0:
    print("stderr tail of the failed run:", *tail, sep="\n  ")
print(f"{time.time() - last:8.2f} s  {stage}")
print(f"{time.time() - t0:8.2f} s  total, rc={p.returncode}, {len(os.listdir(tmp + '/out'))} output files, {nq} queries x {nt} targets")
