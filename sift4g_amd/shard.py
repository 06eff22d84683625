"""Multi-GPU sharding of the scan and the final gather of per-query top-hit lists.

One process per GPU; torch.distributed with backend "nccl" (= RCCL over xGMI on ROCm) on device tensors, "gloo" on
CPU tensors in the tests.  Pairs are independent, so there is NO data-path collective: the only exchange is one
all-gather of the per-query top-K records after the scan.

* query-shard mode: DB replicated in every GPU's HBM, queries split contiguously; gather = concatenation.
* DB-shard mode: every rank scans ALL queries against its slice of the database; gather + K-way merge per query --
  the semantics of dbAlignmentsMerge across database chunks
  (/root/reference/sift4g/src/database_alignment.cpp:97-104) applied across GPUs.
Ordering everywhere: score descending, then global target id ascending.  Unused slots: id -1, score 0.

The gather itself lives BELOW Python: `Comm` wraps the C ABI's s4g_comm_* / s4g_gather_topk (sift4g_amd/csrc/s4g_gather.hip: RCCL
all-gather + pack / unpack / K-way merge kernels on the rank's device), which is what a C++ host calls (tools/gather_demo.cpp).
gather_query_shards / gather_db_shards take `comm=`; without one they go through torch.distributed (gloo in the CPU tests,
where the K-way merge runs in the C ABI's host twin of the merge kernel, s4g_merge_topk_host).
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from ._lib import check, lib

S4G_GATHER_QUERY_SHARD, S4G_GATHER_DB_SHARD = 0, 1


class Comm:
    """One rank of an RCCL communicator owned by libs4g_hip (s4g_comm_create).  `uid`: the 128 bytes of Comm.unique_id(), made by
    one rank and handed to the others; from_process_group() does that over an existing torch.distributed group."""

    def __init__(self, device, rank, world, uid):
        self._h = ctypes.c_void_p()
        self.device, self.rank, self.world = int(device), int(rank), int(world)
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(uid))
        check(lib.s4g_comm_create(ctypes.byref(self._h), self.device, self.rank, self.world, buf), "s4g_comm_create")

    @staticmethod
    def unique_id():
        buf = (ctypes.c_uint8 * 128)()
        check(lib.s4g_comm_unique_id(buf), "s4g_comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_process_group(cls, device, group=None):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        on_dev = dist.get_backend(group) == "nccl"
        t = torch.zeros(129, dtype=torch.uint8)                # the id + "rank 0 has one" (rank 0 broadcasts whatever happened to it)
        err = None
        if rank == 0:
            try:
                t[:128] = torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8)
                t[128] = 1
            except Exception as exc:
                err = exc
        if on_dev:
            t = t.to(torch.device("cuda", device))
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        t = t.cpu()
        if int(t[128]) != 1:
            raise RuntimeError(f"rank 0 could not make a communicator id: {err}" if rank == 0 else "rank 0 could not make a communicator id")
        return cls(device, rank, world, bytes(t[:128].numpy().tobytes()))

    def info(self):
        r, w, d, v = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib.s4g_comm_info(self._h, ctypes.byref(r), ctypes.byref(w), ctypes.byref(d), ctypes.byref(v)), "s4g_comm_info")
        return dict(rank=r.value, world=w.value, device=d.value, rccl_version=v.value)

    def _gather(self, mode, n_queries, ids, scores, counts, id_base, id_map, k_out, stream):
        dev = torch.device("cuda", self.device)
        k = ids.shape[1]
        out_ids = torch.empty((n_queries, k_out), dtype=torch.int32, device=dev)
        out_sc = torch.empty((n_queries, k_out), dtype=torch.int32, device=dev)
        out_cnt = torch.empty((n_queries,), dtype=torch.int32, device=dev)
        cur = torch.cuda.current_stream(dev)
        if stream is None:
            stream = cur
        if counts is None:                       # legal on the torch path: a list's length is its number of used slots
            counts = (ids >= 0).sum(dim=1).to(torch.int32)
        for name, t in (("ids", ids), ("scores", scores), ("counts", counts), ("id_map", id_map)):
            if t is None:
                continue
            if not t.is_cuda or t.device.index != self.device:
                raise ValueError(f"{name} must be a tensor on cuda:{self.device} (got {t.device}); CPU tensors go through gather_db_shards / "
                                 "gather_query_shards of sift4g_amd.shard, which merge on the host")
            if t.dtype != torch.int32:
                raise ValueError(f"{name} must be int32 (got {t.dtype})")
        ids, scores, counts = ids.contiguous(), scores.contiguous(), counts.contiguous()
        if stream != cur:
            # inputs (and the temporaries above) were produced on the current stream and are read by kernels on `stream`:
            # order the two, and keep the caching allocator from recycling the blocks while `stream` still reads them
            stream.wait_stream(cur)
            for t in (ids, scores, counts, id_map, out_ids, out_sc, out_cnt):
                if t is not None:
                    t.record_stream(stream)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() else None)
        check(lib.s4g_gather_topk(self._h, mode, n_queries, k, ptr(ids), ptr(scores), ptr(counts), int(id_base), ptr(id_map), k_out,
                                  ptr(out_ids), ptr(out_sc), ptr(out_cnt), ctypes.c_void_p(stream.cuda_stream)), "s4g_gather_topk")
        return out_ids, out_sc, out_cnt

    def gather_query_shards(self, ids, scores, counts, n_queries_total, stream=None):
        return self._gather(S4G_GATHER_QUERY_SHARD, n_queries_total, ids, scores, counts, 0, None, ids.shape[1], stream)

    def gather_db_shards(self, local_ids, scores, counts, global_ids, k, stream=None):
        if isinstance(global_ids, int):
            return self._gather(S4G_GATHER_DB_SHARD, local_ids.shape[0], local_ids, scores, counts, global_ids, None, k, stream)
        id_map = torch.as_tensor(global_ids).to(torch.device("cuda", self.device)).to(torch.int32).contiguous()
        return self._gather(S4G_GATHER_DB_SHARD, local_ids.shape[0], local_ids, scores, counts, 0, id_map, k, stream)

    def close(self):
        if self._h:
            lib.s4g_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk_lists(ids, scores, k):
    """K-way merge of ORDERED lists on the host -- s4g_merge_topk_host, the host twin of the device merge kernel of
    s4g_gather_topk: ids / scores [lists, Q, M] CPU int32 tensors (global ids, -1 = unused and only at a row's end) -> best k per
    query under (score desc, id asc): (ids [Q, k], scores [Q, k], counts [Q])."""
    lists, q, m = ids.shape
    i_np = np.ascontiguousarray(ids.numpy().astype(np.int32)).view(np.uint32)
    s_np = np.ascontiguousarray(scores.numpy().astype(np.int32))
    if m > 1:
        # the merge ranks by counting and trusts the order of every row: refuse rows that are not in it instead of returning a wrong list
        # (key = score descending, then id ascending; unused slots -- id -1 -- sort last)
        key = (s_np.astype(np.int64) << 32) | (0xFFFFFFFF - i_np.astype(np.int64))
        key[i_np == 0xFFFFFFFF] = -(1 << 62)
        if not (key[..., :-1] >= key[..., 1:]).all():
            raise ValueError("merge_topk_lists: a row is not ordered by (score descending, id ascending) with its unused slots at the end; "
                             "lists in another order (the prefilter's are in id order) must be merged with gather_db_shards(..., ordered=False) / merge_topk")
    o_ids = np.empty((q, k), dtype=np.uint32)
    o_sc = np.empty((q, k), dtype=np.int32)
    o_cnt = np.empty((q,), dtype=np.uint32)
    u32p, i32p = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32)
    check(lib.s4g_merge_topk_host(lists, q, m, i_np.ctypes.data_as(u32p), s_np.ctypes.data_as(i32p), k, o_ids.ctypes.data_as(u32p),
                                  o_sc.ctypes.data_as(i32p), o_cnt.ctypes.data_as(u32p)), "s4g_merge_topk_host")
    return (torch.from_numpy(o_ids.view(np.int32).copy()), torch.from_numpy(o_sc), torch.from_numpy(o_cnt.astype(np.int32)))


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


def gather_query_shards(ids, scores, counts, n_queries_total, group=None, comm=None):
    """query-shard mode: rank r holds the hit lists of queries [b_r, b_{r+1}).  Returns the full
    (ids [Q,K], scores [Q,K], counts [Q]) on every rank.  One all-gather of (2K+1) int32 per query.
    comm: a Comm -- the gather then runs below Python (s4g_gather_topk)."""
    if comm is not None:
        return comm.gather_query_shards(ids, scores, counts, n_queries_total)
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


def gather_db_shards(local_ids, scores, counts, global_ids, k, group=None, comm=None, ordered=True):
    """DB-shard mode: every rank holds, for ALL queries, its local top-k (ids index the rank's own shard).
    global_ids: int tensor mapping local target id -> global id for this rank (ascending), or a plain int = the global id of the
    rank's local target 0 (contiguous shards).  Returns the merged global top-k on every rank.  One all-gather of 2K int32
    per query per rank, then a K-way merge.  comm: a Comm -- gather and merge then run below Python (s4g_gather_topk).
    ordered=False: the rows are NOT in (score desc, id asc) order (the prefilter's lists are in id order): sort-based merge."""
    if comm is not None and ordered:
        return comm.gather_db_shards(local_ids, scores, counts, global_ids, k)
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
    if not out.is_cuda and ordered:                  # host lists (the gloo tests): the C ABI's host twin of the device merge kernel
        return merge_topk_lists(out[:, :, :kk], out[:, :, kk:], k)
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
    g_ids, _, g_counts = gather_db_shards(local, bits, None, global_ids, max_candidates, group, ordered=False)
    key = torch.where(g_ids >= 0, g_ids.to(torch.int64), torch.full_like(g_ids, 1 << 40, dtype=torch.int64))
    out = torch.sort(key, dim=1).values
    return torch.where(out < (1 << 40), out, torch.full_like(out, -1)).to(torch.int32), g_counts
