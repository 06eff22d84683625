"""Multi-process CPU tests (gloo, world size 2) of the sharding + gather logic used on N GPUs: the per-rank
"device" results are emulated with oracle scores, the collectives are the real torch.distributed code path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _topk_ref(scores, k, min_score=1):
    """numpy reference of s4g_topk: (score desc, id asc)"""
    q, n = scores.shape
    ids = np.full((q, k), -1, dtype=np.int32); sc = np.zeros((q, k), dtype=np.int32); cnt = np.zeros(q, dtype=np.int32)
    for i in range(q):
        order = sorted((j for j in range(n) if scores[i, j] >= min_score), key=lambda j: (-scores[i, j], j))[:k]
        ids[i, :len(order)] = order; sc[i, :len(order)] = scores[i, order]; cnt[i] = len(order)
    return ids, sc, cnt


def _worker(rank, world, port, scores, k, mode, lengths, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sift4g_amd import shard
    try:
        nq, nt = scores.shape
        if mode == "query":
            b = shard.shard_bounds(nq, world)
            ids, sc, cnt = _topk_ref(scores[b[rank]:b[rank + 1]], k)
            out = shard.gather_query_shards(torch.from_numpy(ids), torch.from_numpy(sc), torch.from_numpy(cnt), nq)
        else:
            mine = shard.balanced_db_shards(lengths, world)[rank]
            ids, sc, cnt = _topk_ref(scores[:, mine.numpy()], k)
            out = shard.gather_db_shards(torch.from_numpy(ids), torch.from_numpy(sc), torch.from_numpy(cnt), mine, k)
        q.put((rank, [t.numpy() for t in out]))
    finally:
        dist.destroy_process_group()


def _run(world, scores, k, mode, lengths=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scores, k, mode, lengths, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.fixture(scope="module")
def case(oracle_mod, blosum62):
    from conftest import rand_seqs
    rng = np.random.default_rng(41)
    qs = rand_seqs(rng, 7, 30, 90)          # 7 queries: uneven split over 2 ranks
    ts = rand_seqs(rng, 150, 10, 120)
    for j in range(0, 150, 9):
        ts[j] = qs[j % 7][: 20 + j % 40].copy()
    scores = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=4)
    return scores, np.array([len(t) for t in ts])


@pytest.mark.parametrize("k", [5, 40])
def test_query_shard_gather_world2(case, k):
    scores, _ = case
    want = _topk_ref(scores, k)
    res = _run(2, scores, k, "query")
    for r in (0, 1):                        # every rank ends with the full, identical result
        for got, exp in zip(res[r], want):
            assert np.array_equal(got, exp)


@pytest.mark.parametrize("k", [5, 40])
def test_db_shard_merge_world2(case, k):
    scores, lengths = case
    want = _topk_ref(scores, k)
    res = _run(2, scores, k, "db", lengths)
    for r in (0, 1):
        for got, exp in zip(res[r], want):
            assert np.array_equal(got, exp)


def test_single_process_paths_and_helpers(case):
    from sift4g_amd import shard
    scores, lengths = case
    assert shard.shard_bounds(10, 4) == [0, 2, 5, 7, 10]
    parts = shard.balanced_db_shards(lengths, 3)
    allids = torch.sort(torch.cat(parts)).values
    assert allids.tolist() == list(range(len(lengths)))
    tot = [int(lengths[p.numpy()].sum()) for p in parts]
    assert max(tot) - min(tot) <= lengths.max()
    ids, sc, cnt = _topk_ref(scores, 12)
    out = shard.gather_query_shards(torch.from_numpy(ids), torch.from_numpy(sc), torch.from_numpy(cnt), scores.shape[0])
    assert np.array_equal(out[0].numpy(), ids)
    m = shard.merge_topk(torch.from_numpy(np.concatenate([ids, ids], axis=1)), torch.from_numpy(np.concatenate([sc, sc], axis=1)), 24)
    assert (m[2].numpy() == np.minimum(2 * cnt, 24)).all()


# ---- the k-mer prefilter in DB-shard mode: per-rank candidate lists (here from the CPU restatement), merged over gloo ----

def _prefilter_worker(rank, world, port, queries, targets, k, max_candidates, contiguous, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from sift4g_amd import shard
    try:
        if contiguous:
            b = shard.shard_bounds(len(targets), world)
            mine, gmap = list(range(b[rank], b[rank + 1])), b[rank]
        else:
            gmap = shard.balanced_db_shards([len(t) for t in targets], world)[rank]
            mine = [int(i) for i in gmap]
        pf = oracle.Prefilter(queries, k, max_candidates)
        pf.add([targets[i] for i in mine])
        ids, counts, scores, _ = pf.finish()
        out = shard.merge_candidate_shards(torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(counts.astype(np.int64)),
                                           torch.from_numpy(scores), gmap, max_candidates)
        q.put((rank, [t.numpy() for t in out]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("contiguous", [True, False])
def test_prefilter_lists_merge_world2(contiguous):
    import oracle
    from conftest import rand_seqs
    rng = np.random.default_rng(5)
    queries = rand_seqs(rng, 6, 40, 200)
    targets = rand_seqs(rng, 900, 20, 70)                   # short targets: many ties at the cut
    for j in range(0, 900, 31):
        targets[j] = queries[j % 6][: 30 + j % 25].copy()
    k, max_candidates = 4, 25
    whole = oracle.Prefilter(queries, k, max_candidates)
    whole.add(targets)
    w_ids, w_counts, _, cut = whole.finish()
    assert (cut > 0).any()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prefilter_worker, args=(r, 2, port, queries, targets, k, max_candidates, contiguous, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in (0, 1):
        ids, counts = res[r]
        assert counts.tolist() == w_counts.tolist()
        for i in range(len(queries)):
            assert ids[i, :counts[i]].tolist() == w_ids[i, :w_counts[i]].tolist(), (r, i)
            assert (ids[i, counts[i]:] == -1).all()


def _random_ordered_lists(rng, lists, nq, k_in, n_ids=5000, max_score=40):
    """`lists` disjoint shards of a pool of ids with random scores, each shard's per-query list ordered (score desc, id asc), cut at
    k_in, ragged (some lists short, some empty); many score ties"""
    ids = np.full((lists, nq, k_in), -1, dtype=np.int32)
    sc = np.zeros((lists, nq, k_in), dtype=np.int32)
    owner = rng.integers(0, lists, n_ids)
    for q in range(nq):
        s = rng.integers(1, max_score, n_ids)
        for r in range(lists):
            mine = np.nonzero(owner == r)[0]
            if rng.random() < 0.15:
                mine = mine[: rng.integers(0, 4)]
            order = sorted(mine.tolist(), key=lambda j: (-s[j], j))[:k_in]
            ids[r, q, :len(order)] = order
            sc[r, q, :len(order)] = s[order]
    return ids, sc


@pytest.mark.parametrize("lists,k_in,k_out", [(1, 7, 7), (2, 40, 40), (8, 25, 60), (8, 400, 400), (3, 5, 2)])
def test_host_twin_of_the_merge_kernel(lists, k_in, k_out):
    """s4g_merge_topk_host (the host twin of the device merge kernel behind s4g_gather_topk) against the sort-based merge on random
    ordered lists with ties, short and empty rows: ids, scores and counts equal."""
    from sift4g_amd import shard
    rng = np.random.default_rng(lists * 1000 + k_in)
    ids, sc = _random_ordered_lists(rng, lists, 11, k_in)
    got = shard.merge_topk_lists(torch.from_numpy(ids), torch.from_numpy(sc), k_out)
    flat_i = torch.from_numpy(ids).permute(1, 0, 2).reshape(11, lists * k_in)
    flat_s = torch.from_numpy(sc).permute(1, 0, 2).reshape(11, lists * k_in)
    want = shard.merge_topk(flat_i, flat_s, k_out)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
