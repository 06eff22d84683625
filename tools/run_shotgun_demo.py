#!/usr/bin/env python3
"""Writes the configs[1] generator's queries and targets as FASTA and runs build/shotgun_demo on them (both selection
paths).  usage: tools/run_shotgun_demo.py [n_queries=1000] [n_targets=100000] [threads=16]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sift4g_amd import synth
from sift4g_amd.seqio import write_fasta

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
threads = sys.argv[3] if len(sys.argv) > 3 else "16"
tmp = tempfile.mkdtemp()
c = synth.make_config(2, nq, nt)
write_fasta(f"{tmp}/q.fa", [f"q{i}" for i in range(nq)], c["q_codes"], c["q_off"])
write_fasta(f"{tmp}/db.fa", [f"t{i}" for i in range(nt)], c["t_codes"], c["t_off"])
for path in ("0", "1"):
    print(f"== {'shotgunDatabaseEValues' if path == '1' else 'shotgunDatabase (callback)'}", flush=True)
    subprocess.run([f"{ROOT}/build/shotgun_demo", f"{tmp}/q.fa", f"{tmp}/db.fa", "400", "1e-4", threads, path], check=True,
                   env=dict(os.environ, S4G_PROFILE="1"))
