import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sift4g_amd import synth
from sift4g_amd.seqio import write_fasta
nq, nt = 200, 50000
c = synth.make_config(2, nq, nt)
os.makedirs("/tmp/dq/out", exist_ok=True)
write_fasta("/tmp/dq/q.fa", [f"q{i}" for i in range(nq)], c["q_codes"], c["q_off"])
write_fasta("/tmp/dq/db.fa", [f"t{i}" for i in range(nt)], c["t_codes"], c["t_off"])
