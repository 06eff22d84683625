"""Two ranks on the one GPU of the test box (gloo rendezvous, both ranks on cuda:0): the real multi-rank jobs of
sift4g_amd/shard.py -- query-shard and DB-shard, HIP scans + device top-k + gather/merge -- against a single-process scan."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cfg, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sift4g_amd as s4g
        from sift4g_amd import shard
        torch.cuda.set_device(0)
        scorer = s4g.Scorer("BLOSUM_62", 10, 1)
        queries = s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
        # query-shard: DB replicated, queries split
        db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])
        a = [t.cpu().numpy() for t in shard.scan_query_shard(db, queries, scorer, k)]
        # DB-shard: length-balanced slices of the DB, all queries on every rank, streaming top-k + merge
        lengths = np.diff(cfg["t_off"].astype(np.int64))
        mine = shard.balanced_db_shards(lengths, world)[rank].numpy()
        off = np.zeros(len(mine) + 1, dtype=np.uint64); off[1:] = np.cumsum(lengths[mine]).astype(np.uint64)
        codes = np.concatenate([cfg["t_codes"][int(cfg["t_off"][i]):int(cfg["t_off"][i + 1])] for i in mine])
        part = s4g.Database(codes=codes, offsets=off, tile_seqs=256)
        b = [t.cpu().numpy() for t in shard.scan_db_shard(part, mine.astype(np.int64), queries, scorer, k)]
        # the k-mer prefilter on the same slice (device), lists merged across the ranks
        from sift4g_amd import device as s4gd
        qs = [cfg["q_codes"][int(cfg["q_off"][i]):int(cfg["q_off"][i + 1])] for i in range(len(cfg["q_off"]) - 1)]
        pf = s4gd.Prefilter(qs, 5, 15)
        pf.add(codes=codes, offsets=off, first_id=0)
        p_ids, p_counts, p_scores = pf.finish()
        pf.close()
        c = [t.numpy() for t in shard.merge_candidate_shards(torch.from_numpy(p_ids.astype(np.int64)), torch.from_numpy(p_counts.astype(np.int64)),
                                                              torch.from_numpy(p_scores), torch.from_numpy(mine.astype(np.int64)), 15)]
        q.put((rank, a, b, c))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_equal_single_process():
    import sift4g_amd as s4g
    from sift4g_amd import synth
    cfg = synth.make_config(2, n_queries=9, n_targets=1200, homologs_per_query=12)
    cfg = {k: cfg[k] for k in ("q_codes", "q_off", "t_codes", "t_off")}
    k = 40
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])
    queries = s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    full = torch.empty((queries.n, db.n), dtype=torch.int32, device="cuda:0")
    s4g.scan(db, queries, scorer, full)
    ids, sc, cnt = s4g.topk(db, full, k, 1)
    want = (ids.astype(np.int64), sc, cnt.astype(np.int64))
    ctx = mp.get_context("spawn")
    qu = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfg, k, qu)) for r in range(2)]
    for p in procs:
        p.start()
    res = [qu.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the prefilter over the whole database in one process, for the merged lists of the two ranks
    from sift4g_amd import device as s4gd
    qs = [cfg["q_codes"][int(cfg["q_off"][i]):int(cfg["q_off"][i + 1])] for i in range(queries.n)]
    pf = s4gd.Prefilter(qs, 5, 15)
    pf.add(codes=cfg["t_codes"], offsets=cfg["t_off"], first_id=0)
    w_ids, w_counts, _ = pf.finish()
    pf.close()
    assert w_counts.max() == 15                          # the cut binds
    for rank, a, b, c in res:
        assert c[1].tolist() == w_counts.tolist()
        for i in range(queries.n):
            assert c[0][i, :c[1][i]].tolist() == w_ids[i, :w_counts[i]].tolist(), (rank, i)
        for got in (a, b):
            gi = np.where(got[0] < 0, 0xFFFFFFFF, got[0]).astype(np.int64)
            assert np.array_equal(gi, want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2].astype(np.int64), want[2])


@pytest.mark.parametrize("lists,k_in,k_out", [(2, 40, 40), (8, 400, 400), (8, 25, 60), (5, 4096, 4096)])
def test_device_merge_kernel_equals_its_host_twin(lists, k_in, k_out):
    """s4g_merge_topk (the kernel s4g_gather_topk runs behind the RCCL all-gather) on `lists` ordered lists per query in device memory
    against s4g_merge_topk_host on the same lists."""
    import ctypes
    from test_shard import _random_ordered_lists
    from sift4g_amd import shard
    from sift4g_amd._lib import check, lib
    rng = np.random.default_rng(7 * lists + k_in)
    nq = 9
    ids, sc = _random_ordered_lists(rng, lists, nq, k_in, n_ids=3 * k_in * lists // 2 + 50)
    want = shard.merge_topk_lists(torch.from_numpy(ids), torch.from_numpy(sc), k_out)
    d_ids, d_sc = torch.from_numpy(ids).cuda(), torch.from_numpy(sc).cuda()
    o_ids = torch.empty((nq, k_out), dtype=torch.int32, device="cuda")
    o_sc = torch.empty((nq, k_out), dtype=torch.int32, device="cuda")
    o_cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    check(lib.s4g_merge_topk(0, lists, nq, k_in, p(d_ids), p(d_sc), k_out, p(o_ids), p(o_sc), p(o_cnt),
                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "s4g_merge_topk")
    torch.cuda.synchronize()
    assert torch.equal(o_ids.cpu(), want[0]) and torch.equal(o_sc.cpu(), want[1]) and torch.equal(o_cnt.cpu(), want[2])


def test_one_rank_rccl_communicator_below_python():
    """s4g_comm_create + s4g_gather_topk with ONE rank on the box's GPU (RCCL really runs: ncclCommInitRank, ncclAllGather on the
    caller's stream): both modes must hand back what went in -- query-shard the same lists, DB-shard the lists with global ids
    (base offset and id map) -- and agree with the torch.distributed-free path of shard.py."""
    import sift4g_amd as s4g
    from sift4g_amd import shard, synth
    cfg = synth.make_config(2, n_queries=9, n_targets=1200, homologs_per_query=12)
    k = 40
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"])
    queries = s4g.QuerySet(codes=cfg["q_codes"], offsets=cfg["q_off"])
    ids, sc, cnt = s4g.scan_topk(db, queries, scorer, k)
    comm = shard.Comm(0, 0, 1, shard.Comm.unique_id())
    info = comm.info()
    assert (info["rank"], info["world"], info["device"]) == (0, 1, 0) and info["rccl_version"] > 20000
    a = comm.gather_query_shards(ids, sc, cnt, queries.n)
    torch.cuda.synchronize()
    assert torch.equal(a[0], ids) and torch.equal(a[1], sc) and torch.equal(a[2], cnt)
    b = comm.gather_db_shards(ids, sc, cnt, 1000, k)
    torch.cuda.synchronize()
    assert torch.equal(b[0], torch.where(ids >= 0, ids + 1000, ids)) and torch.equal(b[1], sc) and torch.equal(b[2], cnt)
    id_map = torch.arange(db.n, dtype=torch.int64) * 3 + 7
    c = comm.gather_db_shards(ids, sc, cnt, id_map, 25)                      # a shorter output list: the best 25 of the 40
    w = shard.gather_db_shards(ids, sc, cnt, id_map, 25)
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(c, w))
    comm.close()
