"""Multi-GPU sharding of the scan and the final gather of per-query top-hit lists.

One process per GPU; torch.distributed with backend "nccl" (= RCCL over xGMI on ROCm) on device tensors, "gloo" on
CPU tensors in the tests.  Pairs are independent, so there is NO data-path collective: the only exchange is one
all-gather of the per-query top-K records after the scan.

* query-shard mode: DB replicated in every GPU's HBM, queries split contiguously; gather = concatenation.
* DB-shard mode: every rank scans ALL queries against its slice of the database; gather + K-way merge per query --
  the semantics of dbAlignmentsMerge across database chunks
  (/root/reference/sift4g/src/database_alignment.cpp:97-104) applied across GPUs.
Ordering everywhere: score descending, then global target id ascending.  Unused slots: id -1, score 0.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world):
    """contiguous, balanced split of n units over `world` ranks -> list of world+1 offsets"""
    return [(n * r) // world for r in range(world + 1)]


def balanced_db_shards(lengths, world):
    """DB-shard assignment balanced in residues: length-sorted round-robin.
    -> list of index arrays (global target ids, ascending) per rank."""
    order = torch.argsort(torch.as_tensor(lengths), descending=True, stable=True)
    return [torch.sort(order[r::world]).values for r in range(world)]


def _world(group):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _all_gather(out, inp, group):
    """all_gather_into_tensor; with the gloo backend (CPU tests, or two test ranks sharing one GPU) device tensors are
    staged through host memory -- with nccl (= RCCL) they stay on the device."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def gather_query_shards(ids, scores, counts, n_queries_total, group=None):
    """query-shard mode: rank r holds the hit lists of queries [b_r, b_{r+1}).  Returns the full
    (ids [Q,K], scores [Q,K], counts [Q]) on every rank.  One all-gather of (2K+1) int32 per query."""
    world = _world(group)
    if world == 1 and not dist.is_initialized():
        return ids, scores, counts
    bounds = shard_bounds(n_queries_total, world)
    per = max(bounds[r + 1] - bounds[r] for r in range(world))
    k = ids.shape[1]
    packed = torch.zeros((per, 2 * k + 1), dtype=torch.int32, device=ids.device)
    packed[:, :k] = -1
    q = ids.shape[0]
    packed[:q, :k] = ids
    packed[:q, k:2 * k] = scores
    packed[:q, 2 * k] = counts
    out = torch.empty((world * per, 2 * k + 1), dtype=torch.int32, device=ids.device)   # concatenated along dim 0
    _all_gather(out, packed, group)
    out = out.view(world, per, 2 * k + 1)
    rows = torch.cat([out[r, : bounds[r + 1] - bounds[r]] for r in range(world)], dim=0)
    return rows[:, :k].contiguous(), rows[:, k:2 * k].contiguous(), rows[:, 2 * k].contiguous()


def merge_topk(ids, scores, k):
    """K-way merge: ids/scores [Q, M] (any order, id -1 = unused) -> best k per query under (score desc, id asc)."""
    valid = ids >= 0
    key = (scores.to(torch.int64) << 32) | (0xFFFFFFFF - ids.to(torch.int64).clamp(min=0))
    key = torch.where(valid, key, torch.full_like(key, -1))
    top = torch.sort(key, dim=1, descending=True).values[:, :k]
    ok = top >= 0
    out_scores = torch.where(ok, top >> 32, torch.zeros_like(top)).to(torch.int32)
    out_ids = torch.where(ok, 0xFFFFFFFF - (top & 0xFFFFFFFF), torch.full_like(top, -1)).to(torch.int32)
    if out_ids.shape[1] < k:
        pad = k - out_ids.shape[1]
        out_ids = torch.cat([out_ids, torch.full((out_ids.shape[0], pad), -1, dtype=torch.int32, device=ids.device)], dim=1)
        out_scores = torch.cat([out_scores, torch.zeros((out_scores.shape[0], pad), dtype=torch.int32, device=ids.device)], dim=1)
    return out_ids, out_scores, ok.sum(dim=1).to(torch.int32)


def gather_db_shards(local_ids, scores, counts, global_ids, k, group=None):
    """DB-shard mode: every rank holds, for ALL queries, its local top-k (ids index the rank's own shard).
    global_ids: int tensor mapping local target id -> global id for this rank, or a plain int = the global id of the
    rank's local target 0 (contiguous shards).  Returns the merged global top-k on every rank.  One all-gather of 2K int32
    per query per rank, then a K-way merge."""
    world = _world(group)
    valid = local_ids >= 0
    if isinstance(global_ids, int):
        gid = torch.where(valid, local_ids + global_ids, torch.full_like(local_ids, -1))
    else:
        gid = torch.where(valid, global_ids.to(local_ids.device)[local_ids.clamp(min=0).long()].to(torch.int32),
                          torch.full_like(local_ids, -1))
    if world == 1 and not dist.is_initialized():
        return merge_topk(gid, scores, k)
    q, kk = gid.shape
    packed = torch.cat([gid, scores], dim=1).contiguous()
    out = torch.empty((world * q, 2 * kk), dtype=torch.int32, device=gid.device)        # concatenated along dim 0
    _all_gather(out, packed, group)
    out = out.view(world, q, 2 * kk)
    all_ids = out[:, :, :kk].permute(1, 0, 2).reshape(q, world * kk)
    all_sc = out[:, :, kk:].permute(1, 0, 2).reshape(q, world * kk)
    return merge_topk(all_ids, all_sc, k)


def scan_query_shard(db, queries, scorer, k, min_score=1, group=None, rank=None, world=None, scores_buf=None):
    """Whole query-shard job on this rank's GPU: scan my query range against the resident DB, device top-k, RCCL
    gather.  Returns the global (ids, scores, counts) for ALL queries as device tensors."""
    from . import device as dev
    if world is None:
        world = _world(group)
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    bounds = shard_bounds(queries.n, world)
    b, e = bounds[rank], bounds[rank + 1]
    device = torch.device("cuda", db.device)
    if scores_buf is None:
        scores_buf = torch.empty((max(e - b, 1), db.n), dtype=torch.int32, device=device)
    if e > b:
        dev.scan(db, queries, scorer, scores_buf, q_begin=b, q_count=e - b)
        ids, sc, cnt = dev.topk_device(db, scores_buf[: e - b], k, min_score)
    else:
        ids = torch.full((0, k), -1, dtype=torch.int32, device=device)
        sc = torch.zeros((0, k), dtype=torch.int32, device=device)
        cnt = torch.zeros((0,), dtype=torch.int32, device=device)
    return gather_query_shards(ids, sc, cnt, queries.n, group)


def scan_db_shard(db_shard, global_ids, queries, scorer, k, min_score=1, group=None):
    """Whole DB-shard job on this rank's GPU: `db_shard` holds this rank's slice of the database (resident in HBM),
    `global_ids[i]` = global id of its i-th target.  Every rank scans ALL queries with the streaming scan + running
    top-k (no score matrix), then one RCCL all-gather + K-way merge gives the global top-k on every rank."""
    from . import device as dev
    ids, sc, cnt = dev.scan_topk(db_shard, queries, scorer, k, min_score)
    return gather_db_shards(ids, sc, cnt, torch.as_tensor(global_ids), k, group)


def merge_candidate_shards(ids, counts, scores, global_ids, max_candidates, group=None):
    """The k-mer prefilter in DB-shard mode: every rank has filtered its own slice of the database
    (device.Prefilter / s4g_prefilter_*: ids [Q, max_candidates] index the rank's slice, the first counts[q] of a row valid, float
    scores).  One all-gather of the lists and a merge per query under (score descending, global index ascending) give what one
    prefilter over the whole database returns -- the cross-part merge of database_search.cpp:129-150 applied across GPUs.
    global_ids as in gather_db_shards; it must be ascending, so that a rank's own tie rule (lower local index) is the global
    one.  -> (ids [Q, max_candidates] int32 ascending, -1 = unused; counts [Q]) on every rank."""
    ids = torch.as_tensor(ids)
    q, m = ids.shape
    slot = torch.arange(m, device=ids.device).unsqueeze(0)
    valid = slot < torch.as_tensor(counts).to(ids.device).to(torch.int64).unsqueeze(1)
    local = torch.where(valid, ids.to(torch.int64), torch.full((q, m), -1, dtype=torch.int64, device=ids.device)).to(torch.int32)
    bits = torch.as_tensor(scores).to(ids.device).contiguous().view(torch.int32)        # positive floats order as their bit patterns
    bits = torch.where(valid, bits, torch.zeros_like(bits))
    g_ids, _, g_counts = gather_db_shards(local, bits, None, global_ids, max_candidates, group)
    key = torch.where(g_ids >= 0, g_ids.to(torch.int64), torch.full_like(g_ids, 1 << 40, dtype=torch.int64))
    out = torch.sort(key, dim=1).values
    return torch.where(out < (1 << 40), out, torch.full_like(out, -1)).to(torch.int32), g_counts
