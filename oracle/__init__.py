"""ctypes face of oracle/libsw_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
Nothing under sift4g_amd/ imports this package (tests/test_abi.py::test_product_never_touches_the_oracle enforces it).
The oracle is "parity unpinned": see the header of oracle/sw_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_ip = ctypes.POINTER(ctypes.c_int)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsw_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.s4g_oracle_blosum62.argtypes = [_i32p]
        L.s4g_oracle_blosum62.restype = None
        L.s4g_oracle_sw_score.argtypes = [_u8p, ctypes.c_int, _u8p, ctypes.c_int, _i32p, ctypes.c_int, ctypes.c_int]
        L.s4g_oracle_sw_score.restype = ctypes.c_int
        L.s4g_oracle_sw_align.argtypes = [_u8p, ctypes.c_int, _u8p, ctypes.c_int, _i32p, ctypes.c_int, ctypes.c_int,
                                          _ip, _ip, _ip, _ip, _u8p, _ip]
        L.s4g_oracle_sw_align.restype = ctypes.c_int
        L.s4g_oracle_scan.argtypes = [_u8p, _u64p, ctypes.c_uint32, _u8p, _u64p, ctypes.c_uint32,
                                      _i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p]
        L.s4g_oracle_scan.restype = None
        L.s4g_oracle_pairs.argtypes = [_u8p, _u64p, ctypes.c_uint32, _u8p, _u64p, ctypes.c_uint32,
                                       _u32p, _u32p, ctypes.c_uint64,
                                       _i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p]
        L.s4g_oracle_pairs.restype = None
        L.s4g_oracle_simd_available.argtypes = []
        L.s4g_oracle_simd_available.restype = ctypes.c_int
        L.s4g_oracle_scan_simd.argtypes = [_u8p, _u64p, ctypes.c_uint32, _u8p, _u64p, ctypes.c_uint32,
                                           _i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p]
        L.s4g_oracle_scan_simd.restype = ctypes.c_int
        L.s4g_oracle_scan_simd16.argtypes = L.s4g_oracle_scan_simd.argtypes
        L.s4g_oracle_scan_simd16.restype = ctypes.c_int
        L.s4g_oracle_scan_simd8.argtypes = list(L.s4g_oracle_scan_simd.argtypes) + [_u64p]
        L.s4g_oracle_scan_simd8.restype = ctypes.c_int
        _f32p = ctypes.POINTER(ctypes.c_float)
        L.s4g_oracle_prefilter_create.argtypes = [_u8p, _u64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.s4g_oracle_prefilter_create.restype = ctypes.c_void_p
        L.s4g_oracle_prefilter_add.argtypes = [ctypes.c_void_p, _u8p, _u64p, ctypes.c_uint32, ctypes.c_uint32]
        L.s4g_oracle_prefilter_add.restype = None
        L.s4g_oracle_prefilter_finish.argtypes = [ctypes.c_void_p, _u32p, _u32p, _f32p, _f32p]
        L.s4g_oracle_prefilter_finish.restype = None
        L.s4g_oracle_prefilter_lis.argtypes = [ctypes.c_void_p, _u8p, _u64p, ctypes.c_uint32, _i32p]
        L.s4g_oracle_prefilter_lis.restype = None
        L.s4g_oracle_prefilter_destroy.argtypes = [ctypes.c_void_p]
        L.s4g_oracle_prefilter_destroy.restype = None
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def blosum62():
    m = np.zeros(26 * 26, dtype=np.int32)
    lib().s4g_oracle_blosum62(_p(m, _i32p))
    return m.reshape(26, 26)


def encode(seq):
    """letters -> codes 0..25 (letter - 'A'); non-letters dropped (hash.cpp:16-19 alphabet)."""
    b = np.frombuffer(seq.upper().encode("ascii"), dtype=np.uint8)
    b = b[(b >= 65) & (b <= 90)]
    return (b - 65).astype(np.uint8)


def sw_score(q, t, M, gap_open=10, gap_extend=1):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    M = np.ascontiguousarray(M, dtype=np.int32)
    return int(lib().s4g_oracle_sw_score(_p(q, _u8p), len(q), _p(t, _u8p), len(t), _p(M, _i32p), gap_open, gap_extend))


def sw_align(q, t, M, gap_open=10, gap_extend=1):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    M = np.ascontiguousarray(M, dtype=np.int32)
    path = np.zeros(len(q) + len(t) + 1, dtype=np.uint8)
    v = [ctypes.c_int(0) for _ in range(5)]
    s = lib().s4g_oracle_sw_align(_p(q, _u8p), len(q), _p(t, _u8p), len(t), _p(M, _i32p), gap_open, gap_extend,
                                  ctypes.byref(v[0]), ctypes.byref(v[1]), ctypes.byref(v[2]), ctypes.byref(v[3]),
                                  _p(path, _u8p), ctypes.byref(v[4]))
    return dict(score=int(s), q_start=v[0].value, q_end=v[1].value, t_start=v[2].value, t_end=v[3].value,
                path=path[: v[4].value].copy())


def _flat(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    codes = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if len(seqs) and off[-1] else np.zeros(1, np.uint8)
    return np.ascontiguousarray(codes), off


def scan_flat(qcodes, qoff, tcodes, toff, M, gap_open=10, gap_extend=1, threads=1):
    nq, nt = len(qoff) - 1, len(toff) - 1
    out = np.zeros((nq, nt), dtype=np.int32)
    M = np.ascontiguousarray(M, dtype=np.int32)
    if nq and nt:
        lib().s4g_oracle_scan(_p(qcodes, _u8p), _p(qoff, _u64p), nq, _p(tcodes, _u8p), _p(toff, _u64p), nt,
                              _p(M, _i32p), gap_open, gap_extend, threads, _p(out, _i32p))
    return out


def simd_available():
    return bool(lib().s4g_oracle_simd_available())


def scan_simd_flat(qcodes, qoff, tcodes, toff, M, gap_open=10, gap_extend=1, threads=1):
    """AVX2 inter-sequence port (oracle/sw_simd.c), the whole precision ladder (8-bit lanes, int16 for what reached the 8-bit
    saturation mark, scalar beyond int16): the timed CPU baseline.  Returns None without AVX2."""
    nq, nt = len(qoff) - 1, len(toff) - 1
    out = np.zeros((nq, nt), dtype=np.int32)
    M = np.ascontiguousarray(M, dtype=np.int32)
    rc = lib().s4g_oracle_scan_simd(_p(qcodes, _u8p), _p(qoff, _u64p), nq, _p(tcodes, _u8p), _p(toff, _u64p), nt,
                                    _p(M, _i32p), gap_open, gap_extend, threads, _p(out, _i32p))
    return out if rc == 0 else None


def scan_simd16_flat(qcodes, qoff, tcodes, toff, M, gap_open=10, gap_extend=1, threads=1):
    """the int16 tier of the AVX2 port alone (16 targets per register; rounds 1-4's baseline).  Returns None without AVX2."""
    nq, nt = len(qoff) - 1, len(toff) - 1
    out = np.zeros((nq, nt), dtype=np.int32)
    M = np.ascontiguousarray(M, dtype=np.int32)
    rc = lib().s4g_oracle_scan_simd16(_p(qcodes, _u8p), _p(qoff, _u64p), nq, _p(tcodes, _u8p), _p(toff, _u64p), nt,
                                      _p(M, _i32p), gap_open, gap_extend, threads, _p(out, _i32p))
    return out if rc == 0 else None


def scan_simd_ladder_flat(qcodes, qoff, tcodes, toff, M, gap_open=10, gap_extend=1, threads=1):
    """the AVX2 port's precision ladder (8-bit lanes -> int16 -> scalar) -> (scores, pairs the 8-bit tier escalated) or None"""
    nq, nt = len(qoff) - 1, len(toff) - 1
    out = np.zeros((nq, nt), dtype=np.int32)
    esc = np.zeros(1, dtype=np.uint64)
    M = np.ascontiguousarray(M, dtype=np.int32)
    rc = lib().s4g_oracle_scan_simd8(_p(qcodes, _u8p), _p(qoff, _u64p), nq, _p(tcodes, _u8p), _p(toff, _u64p), nt,
                                     _p(M, _i32p), gap_open, gap_extend, threads, _p(out, _i32p), _p(esc, _u64p))
    return (out, int(esc[0])) if rc == 0 else None


def scan(queries, targets, M, gap_open=10, gap_extend=1, threads=1):
    """every query x every target -> int32 [nq, nt]"""
    qc, qo = _flat(queries)
    tc, to = _flat(targets)
    return scan_flat(qc, qo, tc, to, M, gap_open, gap_extend, threads)


def pairs_flat(qcodes, qoff, tcodes, toff, pair_q, pair_t, M, gap_open=10, gap_extend=1, threads=1):
    pair_q = np.ascontiguousarray(pair_q, dtype=np.uint32)
    pair_t = np.ascontiguousarray(pair_t, dtype=np.uint32)
    out = np.zeros(len(pair_q), dtype=np.int32)
    M = np.ascontiguousarray(M, dtype=np.int32)
    if len(pair_q):
        lib().s4g_oracle_pairs(_p(qcodes, _u8p), _p(qoff, _u64p), len(qoff) - 1, _p(tcodes, _u8p), _p(toff, _u64p),
                               len(toff) - 1, _p(pair_q, _u32p), _p(pair_t, _u32p), len(pair_q),
                               _p(M, _i32p), gap_open, gap_extend, threads, _p(out, _i32p))
    return out


class Prefilter:
    """The k-mer prefilter restated (oracle/kmer_prefilter.c; /root/reference/sift4g/src/database_search.cpp:66-280,
    hash.cpp:21-90).  add() takes database chunks in order; finish() -> (ids, counts, scores, cut): rows of max_candidates,
    ids ascending, ties at the cut to the lower database index."""

    def __init__(self, queries, kmer_length=5, max_candidates=5000):
        qc, qo = _flat(queries)
        self.nq, self.max_candidates = len(queries), int(max_candidates)
        self._h = lib().s4g_oracle_prefilter_create(_p(qc, _u8p), _p(qo, _u64p), self.nq, kmer_length, self.max_candidates)
        if not self._h:
            raise ValueError("kmer_length must be 3, 4 or 5")
        self.next_id = 0

    def add(self, targets):
        tc, to = _flat(targets)
        lib().s4g_oracle_prefilter_add(self._h, _p(tc, _u8p), _p(to, _u64p), len(targets), self.next_id)
        self.next_id += len(targets)

    def lis(self, targets):
        """dense [nq, nt] LIS lengths of one chunk (0 = no hit); does not touch the candidate lists"""
        tc, to = _flat(targets)
        out = np.zeros((self.nq, len(targets)), dtype=np.int32)
        if self.nq and len(targets):
            lib().s4g_oracle_prefilter_lis(self._h, _p(tc, _u8p), _p(to, _u64p), len(targets), _p(out, _i32p))
        return out

    def finish(self):
        f32p = ctypes.POINTER(ctypes.c_float)
        ids = np.zeros((self.nq, max(self.max_candidates, 1)), dtype=np.uint32)
        scores = np.zeros(ids.shape, dtype=np.float32)
        counts = np.zeros(max(self.nq, 1), dtype=np.uint32)
        cut = np.zeros(max(self.nq, 1), dtype=np.float32)
        lib().s4g_oracle_prefilter_finish(self._h, _p(ids, _u32p), _p(counts, _u32p), _p(scores, f32p), _p(cut, f32p))
        return ids, counts[: self.nq], scores, cut[: self.nq]

    def close(self):
        if getattr(self, "_h", None) and _LIB is not None:       # module globals are already cleared at interpreter exit
            _LIB.s4g_oracle_prefilter_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()
