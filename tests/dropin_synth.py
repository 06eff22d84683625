"""Shared by tests/test_cpu_pipeline.py and tests/test_gpu_swsharp.py: the reference CLI on a small synthetic workload
(configs[1] generator: 8 queries x 1500 targets with planted homologs), returning the bytes of its alignments.txt."""
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)


def run(dropin, workdir, env=None, nq=8, nt=1500):
    from sift4g_amd import synth
    from sift4g_amd.seqio import write_fasta
    c = synth.make_config(2, nq, nt, homologs_per_query=30)
    write_fasta(os.path.join(workdir, "q.fa"), [f"q{i}" for i in range(nq)], c["q_codes"], c["q_off"])
    write_fasta(os.path.join(workdir, "db.fa"), [f"t{i}" for i in range(nt)], c["t_codes"], c["t_off"])
    out = os.path.join(workdir, "out")
    os.makedirs(out, exist_ok=True)
    r = subprocess.run([dropin, "-q", os.path.join(workdir, "q.fa"), "-d", os.path.join(workdir, "db.fa"), "--cards", "0",
                        "--out", out, "--sub-results", "-t", "4"], env=env, cwd=out, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(os.path.join(out, "alignments.txt"), "rb").read()
