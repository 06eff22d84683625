"""ctypes binding of libs4g_hip.so (include/s4g_hip.h).

There is no fallback: if the shared library is missing or cannot be loaded the
import raises, and every device entry point fails with the library's own error
text when no MI355X is present.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S4G_LIB") or os.path.join(_HERE, "lib", "libs4g_hip.so")   # S4G_LIB: A/B builds of the same ABI

S4G_OK = 0
S4G_ERR_NO_DEVICE = -1
S4G_ERR_INVALID = -2
S4G_ERR_HIP = -3
S4G_ERR_NOMEM = -4
S4G_SCAN_FORCE_I32 = 1
S4G_SCAN_NO_F16 = 2
S4G_SCAN_F16_ARITH = 4
S4G_SCAN_NO_DRIFT = 8
S4G_SCAN_NO_PAIRS = 16
S4G_SCAN_NO_SPEC = 32
S4G_SCAN_BLOCK_PIPE = 64
S4G_SCAN_DIAG_FRAME = 128

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_ip = ctypes.POINTER(ctypes.c_int)
c_vpp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); must list every symbol declared in include/s4g_hip.h
SIGNATURES = {
    "s4g_last_error": (ctypes.c_char_p, []),
    "s4g_abi_version": (ctypes.c_int, []),
    "s4g_device_count": (ctypes.c_int, [c_ip]),
    "s4g_device_check": (ctypes.c_int, [c_ip, ctypes.c_int]),
    "s4g_matrix_by_name": (ctypes.c_int, [ctypes.c_char_p, c_i32p]),
    "s4g_scorer_create": (ctypes.c_int, [c_vpp, c_i32p, ctypes.c_int, ctypes.c_int]),
    "s4g_scorer_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "s4g_db_create": (ctypes.c_int, [c_vpp, ctypes.c_int, c_u8p, c_u64p, ctypes.c_uint32]),
    "s4g_db_create_tiled": (ctypes.c_int, [c_vpp, ctypes.c_int, c_u8p, c_u64p, ctypes.c_uint32, ctypes.c_uint32]),
    "s4g_db_create_scattered": (ctypes.c_int, [c_vpp, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), c_u32p, ctypes.c_uint32]),
    "s4g_db_create_cached": (ctypes.c_int, [c_vpp, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), c_u32p, c_u64p, c_u64p, ctypes.c_uint32]),
    "s4g_chain_cache_info": (ctypes.c_int, [ctypes.c_int, c_u64p, ctypes.c_uint32]),
    "s4g_db_create_from_device": (ctypes.c_int, [c_vpp, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]),
    "s4g_db_tiles": (ctypes.c_int, [ctypes.c_void_p, c_u32p, c_u32p, ctypes.c_uint32]),
    "s4g_db_save": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "s4g_db_load": (ctypes.c_int, [c_vpp, ctypes.c_int, ctypes.c_char_p]),
    "s4g_db_open_streamed": (ctypes.c_int, [c_vpp, ctypes.c_int, ctypes.c_char_p]),
    "s4g_db_stream_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_uint32]),
    "s4g_db_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "s4g_db_info": (ctypes.c_int, [ctypes.c_void_p, c_u32p, c_u64p, c_u64p, c_u64p]),
    "s4g_queries_create": (ctypes.c_int, [c_vpp, ctypes.c_int, c_u8p, c_u64p, ctypes.c_uint32]),
    "s4g_queries_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "s4g_scan_cells": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, c_u64p]),
    "s4g_scan": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "s4g_scan_flags": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]),
    "s4g_scan_host_flags": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.c_void_p, c_i32p, ctypes.c_uint32]),
    "s4g_scan_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                     ctypes.c_void_p, c_i32p]),
    "s4g_score_lists_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_u64p, c_u32p, ctypes.c_void_p, c_i32p]),
    "s4g_score_pairs_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, c_u32p, c_u32p, c_i32p,
                                            ctypes.c_uint32]),
    "s4g_topk": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "s4g_scan_topk": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                     ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_uint32]),
    "s4g_scan_topk_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                          ctypes.c_uint32, c_i32p, c_u32p, c_i32p, c_u32p, ctypes.c_uint32]),
    "s4g_topk_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32,
                                     c_u32p, c_i32p, c_u32p, ctypes.c_void_p]),
    "s4g_traceback_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, c_u32p, c_u32p,
                                          c_i32p, c_i32p, c_u64p, c_u8p]),
    "s4g_traceback_host_flags": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, c_u32p, c_u32p,
                                          c_i32p, c_i32p, c_u64p, c_u8p, ctypes.c_uint32]),
    "s4g_traceback_submit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, c_u32p, c_u32p,
                                            c_i32p, c_u64p, ctypes.POINTER(ctypes.c_void_p)]),
    "s4g_traceback_finish": (ctypes.c_int, [ctypes.c_void_p, c_i32p, ctypes.POINTER(ctypes.c_uint8)]),
    "s4g_warmup": (ctypes.c_int, [ctypes.c_int]),
    "s4g_scan_stats": (ctypes.c_int, [ctypes.c_int, c_u64p, ctypes.c_uint32]),
    "s4g_scan_clock": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_uint32]),
    "s4g_clock_probe": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_uint32]),
    "s4g_scan_plan": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t,
                                     c_u32p, c_u32p, c_u32p, c_u64p]),
    "s4g_prefilter_create": (ctypes.c_int, [c_vpp, ctypes.c_int, c_u8p, c_u64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    "s4g_prefilter_add": (ctypes.c_int, [ctypes.c_void_p, c_u8p, c_u64p, ctypes.c_uint32, ctypes.c_uint32]),
    "s4g_prefilter_add_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]),
    "s4g_prefilter_finish": (ctypes.c_int, [ctypes.c_void_p, c_u32p, c_u32p, ctypes.POINTER(ctypes.c_float)]),
    "s4g_prefilter_stats": (ctypes.c_int, [ctypes.c_void_p, c_u64p, ctypes.c_uint32]),
    "s4g_prefilter_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "s4g_comm_unique_id": (ctypes.c_int, [c_u8p]),
    "s4g_comm_create": (ctypes.c_int, [c_vpp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_u8p]),
    "s4g_comm_info": (ctypes.c_int, [ctypes.c_void_p, c_ip, c_ip, c_ip, c_ip]),
    "s4g_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "s4g_gather_topk": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]),
    "s4g_merge_topk": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "s4g_merge_topk_host": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, c_u32p, c_i32p, ctypes.c_uint32, c_u32p, c_i32p,
                                           c_u32p]),
}

# torch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  If our
# library pulled in /opt/rocm's copy first, a later `import torch` in the same process ends up with two HSA
# runtimes and reports "No HIP GPUs are available".  Loading torch first makes both use one runtime.  torch is
# plumbing here (device buffers, streams, torch.distributed); C++ hosts link /opt/rocm directly and never see it.
try:
    import torch  # noqa: F401
except ImportError:  # pure-ctypes use without torch is fine
    pass

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP library first (`make` or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "sift4g_amd has no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH)
if hasattr(lib, "s4g_is_shim"):
    # tests/cpu_shim/lib/libs4g_hip.so answers the device ABI from the CPU oracle and carries the product library's file name;
    # it marks itself with this symbol so that no S4G_LIB / LD_LIBRARY_PATH setting can put it under the product path
    raise ImportError(f"{LIB_PATH} is the test-only CPU stand-in (tests/cpu_shim), not the HIP library: refusing to load it")
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here = the .so is stale or broken: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


class S4GError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib.s4g_last_error()
        super().__init__(f"{where} failed with code {code}: {msg.decode() if msg else ''}")


def check(code, where):
    if code != S4G_OK:
        raise S4GError(code, where)
