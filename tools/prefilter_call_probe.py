import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from sift4g_amd import synth
import sift4g_amd.device as s4gd
cfg = synth.make_config(cfg=2, n_queries=1000, n_targets=100000)
q_off, t_off = cfg["q_off"], cfg["t_off"]
queries = [cfg["q_codes"][int(q_off[i]):int(q_off[i+1])] for i in range(1000)]
residues = int(t_off[-1])
d_codes = torch.from_numpy(cfg["t_codes"][:residues]).cuda(); d_off = torch.from_numpy(np.asarray(t_off).astype(np.int64)).cuda()
out_bufs = (np.zeros((1000, 5000), np.uint32), np.zeros((1000, 5000), np.float32))
for r in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); p = s4gd.Prefilter(queries, 5, 5000); t1 = time.perf_counter()
    p.add_device(d_codes, d_off, first_id=0); torch.cuda.synchronize(); t2 = time.perf_counter()
    p.finish(scores=False, out=out_bufs); t3 = time.perf_counter()
    p.close(); t4 = time.perf_counter()
    print(f"round {r}: create {1e3*(t1-t0):.3f} add {1e3*(t2-t1):.3f} finish ids {1e3*(t3-t2):.3f} close {1e3*(t4-t3):.3f} ms", file=sys.stderr)
