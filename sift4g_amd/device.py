"""Python face of the C ABI (include/s4g_hip.h): Scorer / Database / QuerySet handles and scan calls.

Names follow the swsharp objects sift4g uses (Scorer, ChainDatabase, Chain query;
/root/reference/sift4g/src/main.cpp:208, database_alignment.cpp:80-86).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import lib, check, c_u8p, c_u32p, c_u64p, c_i32p
from .seqio import flatten


def _ptr(a, t):
    return a.ctypes.data_as(t)


def device_count():
    n = ctypes.c_int(0)
    check(lib.s4g_device_count(ctypes.byref(n)), "s4g_device_count")
    return n.value


def matrix_by_name(name):
    m = np.zeros(26 * 26, dtype=np.int32)
    check(lib.s4g_matrix_by_name(name.encode(), _ptr(m, c_i32p)), "s4g_matrix_by_name")
    return m.reshape(26, 26)


class Scorer:
    """scorerCreateMatrix(&scorer, name, gap_open, gap_extend)  (main.cpp:208)"""

    def __init__(self, matrix="BLOSUM_62", gap_open=10, gap_extend=1):
        table = matrix_by_name(matrix) if isinstance(matrix, str) else np.ascontiguousarray(matrix, dtype=np.int32)
        if table.shape != (26, 26):
            raise ValueError("substitution table must be 26x26")
        self.table = np.ascontiguousarray(table, dtype=np.int32)
        self.gap_open, self.gap_extend = int(gap_open), int(gap_extend)
        self._h = ctypes.c_void_p()
        check(lib.s4g_scorer_create(ctypes.byref(self._h), _ptr(self.table, c_i32p), self.gap_open, self.gap_extend),
              "s4g_scorer_create")

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:       # module globals are already cleared at interpreter exit
            lib.s4g_scorer_destroy(self._h)
            self._h = None


class _SeqSet:
    _create = None
    _destroy = None

    def __init__(self, seqs=None, device=0, codes=None, offsets=None):
        if codes is None:
            codes, offsets = flatten(seqs)
        self.codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n = len(self.offsets) - 1
        self.device = device
        self.lengths = np.diff(self.offsets.astype(np.int64))
        self._h = ctypes.c_void_p()
        create = getattr(self, "_create_override", None) or type(self)._create
        check(create(ctypes.byref(self._h), device, _ptr(self.codes, c_u8p), _ptr(self.offsets, c_u64p), self.n),
              create.__name__)

    def __len__(self):
        return self.n

    def close(self):
        if getattr(self, "_h", None):
            type(self)._destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Database(_SeqSet):
    """chainDatabaseCreate(database, 0, n, cards, cardsLen)  (database_alignment.cpp:80): resident in HBM.
    tile_seqs: sequences per tile of the length-sorted order (0 = library default 2^20)."""
    _create = lib.s4g_db_create
    _destroy = lib.s4g_db_destroy

    def __init__(self, seqs=None, device=0, codes=None, offsets=None, tile_seqs=0):
        if tile_seqs:
            def create(out, dev, c, o, n):
                return lib.s4g_db_create_tiled(out, dev, c, o, n, int(tile_seqs))
            create.__name__ = "s4g_db_create_tiled"
            self._create_override = create
        super().__init__(seqs, device, codes, offsets)

    def save(self, path):
        """write the packed tiles (as resident in HBM) to `path`"""
        check(lib.s4g_db_save(self._h, str(path).encode()), "s4g_db_save")

    @classmethod
    def load(cls, path, device=0):
        """place a database saved with save() on `device` without parsing or packing"""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        check(lib.s4g_db_load(ctypes.byref(self._h), device, str(path).encode()), "s4g_db_load")
        self.device = device
        info = self.info()
        self.n = info["n_seqs"]
        self.codes = self.offsets = None
        self.lengths = None
        return self

    @classmethod
    def open_streamed(cls, path, device=0):
        """a saved database scanned tile by tile out of its file (s4g_db_open_streamed): scan_topk only"""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        check(lib.s4g_db_open_streamed(ctypes.byref(self._h), device, str(path).encode()), "s4g_db_open_streamed")
        self.device = device
        self.n = self.info()["n_seqs"]
        self.codes = self.offsets = None
        self.lengths = None
        return self

    def stream_stats(self):
        out = (ctypes.c_double * 6)()
        check(lib.s4g_db_stream_stats(self._h, out, 6), "s4g_db_stream_stats")
        return dict(bytes_read=out[0], read_s=out[1], wait_s=out[2], tiles=int(out[3]), scan_wall_s=out[4], slot_bytes=int(out[5]))

    @classmethod
    def from_arrays(cls, seqs, device=0):
        """seqs: list of separate uint8 code arrays, one per target (s4g_db_create_scattered: what chainDatabaseCreate passes)."""
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in seqs]
        ptrs = (ctypes.c_void_p * max(len(arrs), 1))(*[a.ctypes.data if a.size else None for a in arrs])
        lens = np.array([a.size for a in arrs], dtype=np.uint32)
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        self.device = device
        check(lib.s4g_db_create_scattered(ctypes.byref(self._h), device, ptrs, _ptr(lens, c_u32p), len(arrs)), "s4g_db_create_scattered")
        self.n = len(arrs)
        self.codes = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.uint8)
        self.offsets = np.concatenate([[0], np.cumsum(lens, dtype=np.uint64)]).astype(np.uint64)
        self.lengths = lens.astype(np.int64)
        return self

    @classmethod
    def from_cached(cls, seqs, tags, addrs, device=0):
        """seqs: list of uint8 code arrays; tags / addrs: uint64 arrays the caller keeps WITH the sequences (zero for a sequence
        the device has not seen; updated in place).  s4g_db_create_cached: every sequence goes to the device once; the database
        serves score_pairs / score_lists / traceback (one wave per pair) only."""
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in seqs]
        ptrs = (ctypes.c_void_p * max(len(arrs), 1))(*[a.ctypes.data if a.size else None for a in arrs])
        lens = np.array([a.size for a in arrs], dtype=np.uint32)
        assert tags.dtype == np.uint64 and addrs.dtype == np.uint64 and len(tags) == len(arrs) == len(addrs)
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        self.device = device
        check(lib.s4g_db_create_cached(ctypes.byref(self._h), device, ptrs, _ptr(lens, c_u32p), _ptr(tags, c_u64p), _ptr(addrs, c_u64p),
                                       len(arrs)), "s4g_db_create_cached")
        self.n = len(arrs)
        self.codes = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.uint8)
        self.offsets = np.concatenate([[0], np.cumsum(lens, dtype=np.uint64)]).astype(np.uint64)
        self.lengths = lens.astype(np.int64)
        return self

    @classmethod
    def from_device(cls, codes, offsets, tile_seqs=0):
        """codes: torch uint8 tensor on the GPU (one byte per residue, sequences back to back); offsets: torch int64 tensor
        [n+1] on the same GPU.  Only the offsets travel to the host; the codes are packed where they are."""
        import torch
        if codes.dtype != torch.uint8 or offsets.dtype != torch.int64 or not codes.is_cuda or codes.device != offsets.device:
            raise ValueError("codes must be a uint8 and offsets an int64 tensor on the same GPU")
        codes, offsets = codes.contiguous(), offsets.contiguous()
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        self.device = codes.device.index or 0
        torch.cuda.current_stream(codes.device).synchronize()          # the library reads the tensors on the null stream
        check(lib.s4g_db_create_from_device(ctypes.byref(self._h), self.device, ctypes.c_void_p(codes.data_ptr()),
                                            ctypes.c_void_p(offsets.data_ptr()), offsets.numel() - 1, int(tile_seqs)),
              "s4g_db_create_from_device")
        self.n = offsets.numel() - 1
        self.codes = None
        self._d_offsets = offsets           # host copies of offsets / lengths only when somebody asks (100 M targets: 1.6 GB)
        return self

    def __getattr__(self, name):            # only reached for attributes that are not set: the lazy ones of from_device()
        if name in ("offsets", "lengths") and "_d_offsets" in self.__dict__:
            off = self.__dict__["_d_offsets"].cpu().numpy().view(np.uint64)
            self.offsets = off
            self.lengths = np.diff(off).astype(np.uint32)
            return self.__dict__[name]
        raise AttributeError(name)

    def tiles(self):
        n = ctypes.c_uint32()
        check(lib.s4g_db_tiles(self._h, ctypes.byref(n), None, 0), "s4g_db_tiles")
        arr = (ctypes.c_uint32 * max(n.value, 1))()
        check(lib.s4g_db_tiles(self._h, ctypes.byref(n), arr, n.value), "s4g_db_tiles")
        return list(arr)[: n.value]

    def info(self):
        n = ctypes.c_uint32(); r = ctypes.c_uint64(); p = ctypes.c_uint64(); b = ctypes.c_uint64()
        check(lib.s4g_db_info(self._h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(p), ctypes.byref(b)), "s4g_db_info")
        return dict(n_seqs=n.value, residues=r.value, padded_residues=p.value, hbm_bytes=b.value)


class QuerySet(_SeqSet):
    _create = lib.s4g_queries_create
    _destroy = lib.s4g_queries_destroy


def scan_cells(db, queries, q_begin=0, q_count=None):
    q_count = queries.n - q_begin if q_count is None else q_count
    c = ctypes.c_uint64()
    check(lib.s4g_scan_cells(db._h, queries._h, q_begin, q_count, ctypes.byref(c)), "s4g_scan_cells")
    return c.value


def scan_plan(db, queries, scorer):
    name = ctypes.create_string_buffer(128)
    g = ctypes.c_uint32(); b = ctypes.c_uint32(); s = ctypes.c_uint32(); w = ctypes.c_uint64()
    check(lib.s4g_scan_plan(db._h, queries._h, scorer._h, name, 128, ctypes.byref(g), ctypes.byref(b), ctypes.byref(s),
                            ctypes.byref(w)), "s4g_scan_plan")
    return dict(kernel=name.value.decode(), grid_blocks=g.value, block_threads=b.value, strip_rows=s.value,
                workspace_bytes=w.value)


def scan_stats(device=0):
    """what the last scan call on `device` did (s4g_scan_stats): queries per tier, speculative routing, rechecked entries, pairs"""
    out = np.zeros(11, dtype=np.uint64)
    check(lib.s4g_scan_stats(device, _ptr(out, c_u64p), 11), "s4g_scan_stats")
    return dict(tier_drift=int(out[0]), tier_m3=int(out[1]), tier_i16=int(out[2]), tier_i32=int(out[3]),
                speculative=int(out[4]), rechecked=int(out[5]), pair_items=int(out[6]), list_pairs=int(out[7]),
                coop_scan=int(out[8]), coop_trace=int(out[9]), pair_pk=int(out[10]))


def chain_cache_info(device=0):
    out = np.zeros(7, dtype=np.uint64)
    check(lib.s4g_chain_cache_info(device, _ptr(out, c_u64p), 7), "s4g_chain_cache_info")
    return dict(chains=int(out[0]), bytes=int(out[1]), hits=int(out[2]), misses=int(out[3]), generation=int(out[4]),
                databases=int(out[5]), alive=int(out[6]))


def scan_clock(device=0):
    """shader clock and duration of the longest scan launch of the last scan call on `device`, sampled inside the kernel
    (s4g_scan_clock)"""
    out = (ctypes.c_double * 10)()
    check(lib.s4g_scan_clock(device, out, 10), "s4g_scan_clock")
    return dict(launches=int(out[0]), shader_mhz=out[1], shader_mhz_min=out[2], shader_mhz_max=out[3], blocks=int(out[4]),
                kernel_ms=out[5], tier=("drift", "m3", "i16", "i32")[int(out[6])] if out[0] else None, shader_clocks=out[7],
                block_ms_mean=out[8], block_ms_min=out[9])


def clock_probe(device=0):
    """shader clock under vector-ALU load and the measured packed-int16 issue rate (s4g_clock_probe)"""
    out = (ctypes.c_double * 5)()
    check(lib.s4g_clock_probe(device, out, 5), "s4g_clock_probe")
    return dict(shader_mhz=out[0], pk_add_T_lane_ops=out[1], cus=int(out[2]), nominal_mhz=out[3], probe_ms=out[4])


def _flags(force_i32=False, no_f16=False, f16_arith=False, no_drift=False, no_pairs=False, no_spec=False, block_pipe=False, diag_frame=False):
    return ((_lib.S4G_SCAN_DIAG_FRAME if diag_frame else 0) | (_lib.S4G_SCAN_FORCE_I32 if force_i32 else 0) | (_lib.S4G_SCAN_NO_F16 if no_f16 else 0) | (_lib.S4G_SCAN_BLOCK_PIPE if block_pipe else 0) |
            (_lib.S4G_SCAN_F16_ARITH if f16_arith else 0) | (_lib.S4G_SCAN_NO_DRIFT if no_drift else 0) |
            (_lib.S4G_SCAN_NO_PAIRS if no_pairs else 0) | (_lib.S4G_SCAN_NO_SPEC if no_spec else 0))


def scan(db, queries, scorer, out, q_begin=0, q_count=None, stream=None, force_i32=False, no_f16=False, f16_arith=False, no_drift=False,
         no_pairs=False, no_spec=False, block_pipe=False, diag_frame=False):
    """Asynchronous scan into a device tensor `out` (torch int32 [q_count, n_seqs], on db.device).

    stream: a torch.cuda.Stream (default: torch's current stream on that device)."""
    import torch
    q_count = queries.n - q_begin if q_count is None else q_count
    if out.dtype != torch.int32 or not out.is_contiguous() or out.numel() < q_count * db.n or not out.is_cuda:
        raise ValueError("out must be a contiguous CUDA int32 tensor of at least q_count*n_seqs elements")
    if stream is None:
        stream = torch.cuda.current_stream(out.device)
    check(lib.s4g_scan_flags(db._h, queries._h, q_begin, q_count, scorer._h, ctypes.c_void_p(out.data_ptr()),
                             ctypes.c_void_p(stream.cuda_stream), _flags(force_i32, no_f16, f16_arith, no_drift, no_pairs, no_spec, block_pipe, diag_frame)), "s4g_scan_flags")
    return out


def scan_host(db, queries, scorer, q_begin=0, q_count=None, force_i32=False, no_f16=False, f16_arith=False, no_drift=False,
              no_pairs=False, no_spec=False, block_pipe=False, diag_frame=False):
    """Synchronous scan -> numpy int32 [q_count, n_seqs]."""
    q_count = queries.n - q_begin if q_count is None else q_count
    out = np.zeros((q_count, db.n), dtype=np.int32)
    check(lib.s4g_scan_host_flags(db._h, queries._h, q_begin, q_count, scorer._h, _ptr(out, c_i32p),
                                  _flags(force_i32, no_f16, f16_arith, no_drift, no_pairs, no_spec, block_pipe, diag_frame)), "s4g_scan_host_flags")
    return out


def score_lists(db, queries, scorer, cand_offsets, cand_ids):
    cand_offsets = np.ascontiguousarray(cand_offsets, dtype=np.uint64)
    cand_ids = np.ascontiguousarray(cand_ids, dtype=np.uint32)
    out = np.zeros(max(int(cand_offsets[-1]), 1), dtype=np.int32)
    check(lib.s4g_score_lists_host(db._h, queries._h, _ptr(cand_offsets, c_u64p), _ptr(cand_ids, c_u32p), scorer._h,
                                   _ptr(out, c_i32p)), "s4g_score_lists_host")
    return out[: int(cand_offsets[-1])]


def score_pairs(db, queries, scorer, pair_q, pair_t, route=None):
    """scores of explicit (query, target) pairs; route = None (cost model) | "wave" (one wave per pair) | "scan" (full scan +
    column pick) | "lists" (candidates gathered into 128-lane groups for the packed list kernel, long targets as waves)."""
    pair_q = np.ascontiguousarray(pair_q, dtype=np.uint32)
    pair_t = np.ascontiguousarray(pair_t, dtype=np.uint32)
    if pair_q.shape != pair_t.shape or pair_q.ndim != 1:
        raise ValueError("pair_q and pair_t must be 1-D arrays of the same length")
    out = np.zeros(max(len(pair_q), 1), dtype=np.int32)
    flags = (4 | (int(route[6:]) << 8)) if isinstance(route, str) and route.startswith("lists:") else {None: 0, "wave": 1, "scan": 2, "lists": 4}[route]
    check(lib.s4g_score_pairs_host(db._h, queries._h, scorer._h, len(pair_q), _ptr(pair_q, c_u32p), _ptr(pair_t, c_u32p),
                                   _ptr(out, c_i32p), flags), "s4g_score_pairs_host")
    return out[: len(pair_q)]


def scan_topk(db, queries, scorer, k, min_score=1, q_begin=0, q_count=None, stream=None, force_i32=False, no_f16=False,
              f16_arith=False, no_drift=False, no_pairs=False, no_spec=False, diag_frame=False):
    """Streaming scan + running top-k (no score matrix): -> device tensors (ids int32 [q,k] (-1 unused), scores [q,k], counts [q])."""
    import torch
    q_count = queries.n - q_begin if q_count is None else q_count
    device = torch.device("cuda", db.device)
    ids = torch.empty((q_count, k), dtype=torch.int32, device=device)
    sc = torch.empty((q_count, k), dtype=torch.int32, device=device)
    cnt = torch.empty((q_count,), dtype=torch.int32, device=device)
    if stream is None:
        stream = torch.cuda.current_stream(device)
    check(lib.s4g_scan_topk(db._h, queries._h, q_begin, q_count, scorer._h, k, int(min_score), ctypes.c_void_p(ids.data_ptr()),
                            ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(stream.cuda_stream),
                            _flags(force_i32, no_f16, f16_arith, no_drift, no_pairs, no_spec, diag_frame=diag_frame)), "s4g_scan_topk")
    return ids, sc, cnt


def topk_device(db, scores, k, min_score=1, stream=None):
    """Per-query top-k kept on the device: -> (ids int32 [q,k] (-1 = unused), scores int32 [q,k], counts int32 [q]) torch tensors.
    This is the form the RCCL gather consumes (sift4g_amd/shard.py)."""
    import torch
    q = scores.shape[0]
    ids = torch.empty((q, k), dtype=torch.int32, device=scores.device)
    sc = torch.empty((q, k), dtype=torch.int32, device=scores.device)
    cnt = torch.empty((q,), dtype=torch.int32, device=scores.device)
    if stream is None:
        stream = torch.cuda.current_stream(scores.device)
    check(lib.s4g_topk(db._h, ctypes.c_void_p(scores.data_ptr()), q, k, int(min_score), ctypes.c_void_p(ids.data_ptr()),
                       ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(stream.cuda_stream)),
          "s4g_topk")
    return ids, sc, cnt


def topk(db, scores, k, min_score=1, stream=None):
    """Per-query top-k of a device score tensor [q, n_seqs] -> (ids uint32 [q,k], scores int32 [q,k], counts uint32 [q])."""
    import torch
    q = scores.shape[0]
    ids = np.zeros((q, k), dtype=np.uint32)
    sc = np.zeros((q, k), dtype=np.int32)
    cnt = np.zeros(q, dtype=np.uint32)
    if stream is None:
        stream = torch.cuda.current_stream(scores.device)
    check(lib.s4g_topk_host(db._h, ctypes.c_void_p(scores.data_ptr()), q, k, int(min_score), _ptr(ids, c_u32p),
                            _ptr(sc, c_i32p), _ptr(cnt, c_u32p), ctypes.c_void_p(stream.cuda_stream)), "s4g_topk_host")
    return ids, sc, cnt


def traceback(db, queries, scorer, pair_q, pair_t, pair_score, route=None):
    """Device traceback of (query, target) pairs with known scores -> list of dicts(score, q_start, q_end, t_start, t_end, path)
    (path: uint8 array of swsharp move codes 1 DIAG / 2 LEFT / 3 UP, start to end).
    route = None / "wave" (one wave per pair) | "scan" (the 64-pairs-per-wave strip kernels)."""
    pair_q = np.ascontiguousarray(pair_q, dtype=np.uint32)
    pair_t = np.ascontiguousarray(pair_t, dtype=np.uint32)
    pair_score = np.ascontiguousarray(pair_score, dtype=np.int32)
    n = len(pair_q)
    # (out-of-range ids are reported by the library, not by numpy: clip only for sizing the path buffers)
    cap = (np.take(queries.lengths, pair_q, mode="clip").astype(np.uint64) +
           np.take(db.lengths, pair_t, mode="clip").astype(np.uint64)) if n else np.zeros(0, np.uint64)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(cap)
    out = np.zeros((max(n, 1), 5), dtype=np.int32)
    paths = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
    check(lib.s4g_traceback_host_flags(db._h, queries._h, scorer._h, n, _ptr(pair_q, c_u32p), _ptr(pair_t, c_u32p),
                                       _ptr(pair_score, c_i32p), _ptr(out, c_i32p), _ptr(off, c_u64p), _ptr(paths, c_u8p),
                                       {None: 0, "wave": 1, "scan": 2}[route]), "s4g_traceback_host")
    return [dict(score=int(pair_score[i]), q_start=int(out[i, 0]), q_end=int(out[i, 1]), t_start=int(out[i, 2]),
                 t_end=int(out[i, 3]), path=paths[int(off[i]):int(off[i]) + int(out[i, 4])].copy()) for i in range(n)]


class TraceJob:
    """s4g_traceback_submit / s4g_traceback_finish (include/s4g_hip.h): the traceback of `traceback` in two halves.  The job is
    queued at construction; result() waits for it and returns what traceback() returns, once."""

    def __init__(self, db, queries, scorer, pair_q, pair_t, pair_score):
        self.pair_q = np.ascontiguousarray(pair_q, dtype=np.uint32)
        self.pair_t = np.ascontiguousarray(pair_t, dtype=np.uint32)
        self.pair_score = np.ascontiguousarray(pair_score, dtype=np.int32)
        n = self.n = len(self.pair_q)
        cap = (np.take(queries.lengths, self.pair_q, mode="clip").astype(np.uint64) +
               np.take(db.lengths, self.pair_t, mode="clip").astype(np.uint64)) if n else np.zeros(0, np.uint64)
        self.off = np.zeros(n + 1, dtype=np.uint64)
        self.off[1:] = np.cumsum(cap)
        self._queries = queries                       # the query set must outlive the job (header); the database need not
        self._h = ctypes.c_void_p()
        check(lib.s4g_traceback_submit(db._h, queries._h, scorer._h, n, _ptr(self.pair_q, c_u32p), _ptr(self.pair_t, c_u32p),
                                       _ptr(self.pair_score, c_i32p), _ptr(self.off, c_u64p), ctypes.byref(self._h)),
              "s4g_traceback_submit")

    def result(self):
        if self._h is None:
            raise RuntimeError("a trace job is finished exactly once")
        n, off = self.n, self.off
        out = np.zeros((max(n, 1), 5), dtype=np.int32)
        paths = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
        h, self._h = self._h, None
        check(lib.s4g_traceback_finish(h, _ptr(out, c_i32p), _ptr(paths, c_u8p)), "s4g_traceback_finish")
        self._queries = None
        return [dict(score=int(self.pair_score[i]), q_start=int(out[i, 0]), q_end=int(out[i, 1]), t_start=int(out[i, 2]),
                     t_end=int(out[i, 3]), path=paths[int(off[i]):int(off[i]) + int(out[i, 4])].copy()) for i in range(n)]


class Prefilter:
    """searchDatabase's compute on the device (database_search.cpp:66-253, hash.cpp:21-90): the queries' k-mer index, then
    database chunks in ascending id order through add(); finish() -> (ids [nq, max_candidates], counts [nq], scores) with the
    ids of a row ascending.  Ties at the cut go to the lower database index (include/s4g_hip.h)."""

    def __init__(self, queries, kmer_length=5, max_candidates=5000, device=0):
        codes, offsets = flatten(queries)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.nq, self.max_candidates, self.next_id = len(offsets) - 1, int(max_candidates), 0
        self._h = ctypes.c_void_p()
        check(lib.s4g_prefilter_create(ctypes.byref(self._h), device, _ptr(codes, c_u8p), _ptr(offsets, c_u64p), self.nq,
                                       int(kmer_length), self.max_candidates), "s4g_prefilter_create")

    def add(self, targets=None, codes=None, offsets=None, first_id=None):
        if codes is None:
            codes, offsets = flatten(targets)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        first = self.next_id if first_id is None else int(first_id)
        check(lib.s4g_prefilter_add(self._h, _ptr(codes, c_u8p), _ptr(offsets, c_u64p), n, first), "s4g_prefilter_add")
        self.next_id = first + n

    def add_device(self, codes, offsets, first_id=None):
        """codes: uint8 device tensor, offsets: int64 device tensor of n + 1 entries (relative to codes)"""
        n = offsets.numel() - 1
        first = self.next_id if first_id is None else int(first_id)
        check(lib.s4g_prefilter_add_device(self._h, codes.data_ptr(), offsets.data_ptr(), n, first), "s4g_prefilter_add_device")
        self.next_id = first + n

    def finish(self, scores=True, out=None):
        """-> (ids [nq, max_candidates] uint32, counts [nq], scores float32 or None).  scores=False: ids only, as searchDatabase
        returns them (database_search.cpp:166-178) and the drop-in asks for them; out: a (ids, scores) pair of arrays to fill (a caller
        that keeps its buffers pays their first-touch page faults once)."""
        shape = (max(self.nq, 1), self.max_candidates)
        ids = out[0] if out is not None else np.zeros(shape, dtype=np.uint32)
        sc = (out[1] if out is not None else np.zeros(shape, dtype=np.float32)) if scores else None
        counts = np.zeros(max(self.nq, 1), dtype=np.uint32)
        check(lib.s4g_prefilter_finish(self._h, _ptr(ids, c_u32p), _ptr(counts, c_u32p),
                                       _ptr(sc, ctypes.POINTER(ctypes.c_float)) if scores else None), "s4g_prefilter_finish")
        return ids[: self.nq], counts[: self.nq], (sc[: self.nq] if scores else None)

    def stats(self):
        out = np.zeros(7, dtype=np.uint64)
        check(lib.s4g_prefilter_stats(self._h, _ptr(out, c_u64p), 7), "s4g_prefilter_stats")
        return dict(hits=int(out[0]), pairs=int(out[1]), long_pairs=int(out[2]), chunks=int(out[3]), fused=bool(out[4]),
                    general_path_targets=int(out[5]), repeated_chunks=int(out[6]))

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.s4g_prefilter_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()
