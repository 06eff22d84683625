"""What needs MORE than one GPU -- and says so when the box has only one.

The driver's test box has one MI355X; its scaling box (when one exists) has eight.  Every test here that needs two devices SKIPS
with the device count in the reason on a one-GPU box and RUNS on a multi-GPU box, so the first real multi-GPU run validates
correctness, not only speed:

  * build/gather_demo (tools/gather_demo.cpp): N processes, plain C++ over the C ABI -- scan, s4g_comm_create, s4g_gather_topk --
    self-checked against one GPU scanning everything; one rank always, two ranks on two devices when there are two;
  * libswsharp's in-process multi-card path (`--cards 01`: one host thread and one database replica per card,
    /root/reference/sift4g/src/main.cpp:254-262 -> database_alignment.cpp:83-86) on DISTINCT devices;
  * `python bench.py --gpus 2` over real RCCL: rccl_ranks == 2, not emulated, gather through the C ABI.
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, rand_seqs

pytestmark = pytest.mark.gpu

DEMO = os.path.join(ROOT, "build", "gather_demo")


def _need_two_gpus():
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"needs 2 GPUs, this box has {n}: the multi-device path is covered by one-rank RCCL + emulated ranks only")
    return n


def _demo(*args, timeout=600):
    assert os.path.exists(DEMO), "build/gather_demo missing: run `make` (or __graft_entry__.build())"
    p = subprocess.run([DEMO, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr[-2000:])
    last = [l for l in p.stdout.splitlines() if l.startswith("gather_demo")][-1]
    assert ": OK (0 mismatches" in last, last
    return last, p.stderr


@pytest.mark.parametrize("mode", ["query-shard", "db-shard"])
def test_cpp_host_over_the_c_abi_one_rank(mode):
    """the C++ host program with ONE rank: RCCL communicator, scan, gather, self-check -- no Python below the test harness"""
    last, err = _demo("--ranks", "1", "--mode", mode, "--queries", "48", "--targets", "15000", "--k", "50")
    assert f"gather_demo {mode} ranks=1" in last and "rank 0 of 1" in err


@pytest.mark.parametrize("mode", ["query-shard", "db-shard"])
def test_cpp_host_over_the_c_abi_two_ranks_two_gpus(mode):
    """two processes, two devices, RCCL over xGMI: the gathered lists equal one GPU scanning everything"""
    _need_two_gpus()
    last, err = _demo("--ranks", "2", "--mode", mode, "--queries", "51", "--targets", "30001", "--k", "50")
    assert f"gather_demo {mode} ranks=2" in last and "rank 1 of 2, device 1" in err


def test_shotgun_database_on_two_distinct_cards(oracle_mod, blosum62):
    """libswsharp's multi-card path on two DIFFERENT devices (cards = {0, 1}): a replica of the database per card, the queries split
    over two host threads, results per query equal to selection on oracle scores"""
    _need_two_gpus()
    import swsharp_ctypes as sw
    L = sw.lib
    ci, pvp = ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)
    rng = np.random.default_rng(61)
    qs = rand_seqs(rng, 12, 80, 300)
    ts = rand_seqs(rng, 400, 30, 400)
    for k in range(50):
        m = qs[k % 12].copy(); idx = rng.choice(len(m), size=len(m) // 3, replace=False)
        m[idx] = rand_seqs(rng, 1, len(idx), len(idx))[0]; ts[k * 7] = m
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    s = sw.scorer(10, 1)
    tch = [sw.chain(f"t{i}", t) for i, t in enumerate(ts)]
    qch = [sw.chain(f"q{i}", q) for i, q in enumerate(qs)]
    tarr, qarr = sw.chain_array(tch), sw.chain_array(qch)
    ev = L.createEValueParams(int(sum(len(t) for t in ts)), s)
    cb = sw.evalue_callback()
    cards = (ci * 2)(0, 1)
    cdb = L.chainDatabaseCreate(tarr, 0, len(tch), cards, 2)
    outs, lens = ctypes.POINTER(pvp)(), ctypes.POINTER(ci)()
    L.shotgunDatabase(ctypes.byref(outs), ctypes.byref(lens), 0, qarr, len(qch), cdb, s, 10, cb, ev, 1e-3, None, 0, cards, 2, None)
    from test_gpu_swsharp import _hits
    for i in range(len(qs)):
        ref, nr = pvp(), ci()
        sc = (ci * len(ts))(*[int(x) for x in exp[i]])
        L.dbAlignmentsFromScores(ctypes.byref(ref), ctypes.byref(nr), 0, qch[i], tarr, len(tch), sc, s, 10, cb, ev, 1e-3)
        assert lens[i] == nr.value and _hits(outs[i], lens[i]) == _hits(ref, nr.value)
    L.deleteShotgunDatabase(outs, lens, len(qs))
    L.chainDatabaseDelete(cdb)
    L.deleteEValueParams(ev); L.scorerDelete(s)


def test_bench_two_gpus_over_real_rccl():
    """`python bench.py --gpus 2` on two devices: two RCCL ranks (not the gloo emulation), the gather below Python, self-check clean,
    in both sharding modes"""
    _need_two_gpus()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra in (["--queries", "64", "--targets", "20000"], ["--mode", "db-shard", "--queries", "8", "--targets", "60000"]):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *extra],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads(p.stdout.strip().splitlines()[-1])
        assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "RCCL (nccl)" and not d["emulated"]
        assert d["gather"].startswith("s4g_gather_topk") and d["selfcheck"]["mismatches"] == 0
        assert d["phases"]["gather_ms"]["max"] > 0
