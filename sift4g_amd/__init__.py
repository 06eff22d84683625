"""sift4g_amd -- MI355X-native Smith-Waterman database scoring for the sift4g pipeline.

The product is the HIP library `lib/libs4g_hip.so` (C ABI: include/s4g_hip.h) and the
swsharp-compatible C++ host library above it.  This package is the thin Python face
used by tests, bench.py and the multi-GPU driver: ctypes for the C ABI, torch only for
device buffers, streams and torch.distributed.
"""
from . import _lib  # noqa: F401  (raises if the HIP library is not built: no fallback)
from .device import Scorer, Database, QuerySet, device_count, scan, scan_host, topk, topk_device, scan_topk, score_lists, score_pairs, traceback, TraceJob  # noqa: F401
from .seqio import encode, decode, read_fasta  # noqa: F401
