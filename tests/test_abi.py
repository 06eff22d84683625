"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the header declares,
and fails loudly (no fallback) when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "s4g_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(s4g_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sift4g_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 17
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/s4g_hip.h but not exported by libs4g_hip.so"
    assert sorted(_lib.SIGNATURES) == declared, "sift4g_amd/_lib.py SIGNATURES out of sync with include/s4g_hip.h"
    assert _lib.lib.s4g_abi_version() == 6


def test_header_cites_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "s4g_hip.h")).read()
    for cite in ("database_alignment.cpp:83-86", "database_alignment.cpp:80", "main.cpp:208", "main.cpp:183-186"):
        assert cite in text


def test_builtin_matrix_equals_oracle_table(blosum62):
    from sift4g_amd.device import matrix_by_name
    assert np.array_equal(matrix_by_name("BLOSUM_62"), blosum62)
    from sift4g_amd._lib import S4GError
    with pytest.raises(S4GError):
        matrix_by_name("BLOSUM_13")


def test_scorer_validation():
    from sift4g_amd import Scorer
    from sift4g_amd._lib import S4GError
    Scorer("BLOSUM_62", 10, 1)
    with pytest.raises(S4GError):
        Scorer("BLOSUM_62", 1, 2)        # extend > open (main.cpp:297-303)
    bad = np.zeros((26, 26), dtype=np.int32); bad[0, 0] = 1000
    with pytest.raises(S4GError):
        Scorer(bad, 10, 1)


def test_device_check_accepts_zero_cards():
    # cudaCheckCards must pass for 0 cards (main.cpp:78-79,186)
    from sift4g_amd._lib import lib
    assert lib.s4g_device_check(None, 0) == 1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_fails_loudly():
    import sift4g_amd as s
    from sift4g_amd._lib import S4GError, S4G_ERR_NO_DEVICE
    with pytest.raises(S4GError) as e:
        s.Database([np.zeros(4, dtype=np.uint8)])
    assert e.value.code == S4G_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(S4GError):
        s.QuerySet([np.zeros(4, dtype=np.uint8)])
    from sift4g_amd import device as s4gd
    with pytest.raises(S4GError) as e:                      # the k-mer prefilter has no CPU path either
        s4gd.Prefilter([np.zeros(8, dtype=np.uint8)], 5, 10)
    assert e.value.code == S4G_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under sift4g_amd/ (or the C++ sources) may reference it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "sift4g_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", ".c")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"\bimport oracle\b|\bfrom oracle\b|sw_oracle|libsw_oracle|s4g_oracle_", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad



def test_the_cpu_stand_in_cannot_be_loaded_as_the_product_library():
    """tests/cpu_shim/lib/libs4g_hip.so answers the device ABI from the oracle under the product library's file name; the
    Python face refuses it (s4g_is_shim), whatever S4G_LIB says."""
    import subprocess, sys
    shim = os.path.join(ROOT, "tests", "cpu_shim")
    subprocess.run(["make", "-s", "-C", shim], check=True, capture_output=True, timeout=300)
    r = subprocess.run([sys.executable, "-c", "import sift4g_amd"], cwd=ROOT, capture_output=True, text=True,
                       env=dict(os.environ, S4G_LIB=os.path.join(shim, "lib", "libs4g_hip.so")))
    assert r.returncode != 0 and "stand-in" in r.stderr
    raw = ctypes.CDLL(os.path.join(ROOT, "sift4g_amd", "lib", "libs4g_hip.so"))
    assert not hasattr(raw, "s4g_is_shim")


def test_missing_rccl_is_an_error_code_not_a_crash():
    """with no librccl to open, s4g_comm_unique_id must return S4G_ERR_NO_DEVICE with the loader's message --
    round 4 asked dlerror() twice, got NULL the second time and built a std::string from it (SIGSEGV inside call_once).
    S4G_RCCL_LIBRARY points the loader at a file that does not exist; a child process, since the loader state is per process."""
    import subprocess, sys
    code = ("import ctypes, sys; L = ctypes.CDLL(sys.argv[1]); L.s4g_last_error.restype = ctypes.c_char_p; "
            "buf = (ctypes.c_uint8 * 128)(); rc = L.s4g_comm_unique_id(buf); print(rc, L.s4g_last_error().decode())")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "sift4g_amd", "lib", "libs4g_hip.so")], capture_output=True, text=True,
                       timeout=120, env=dict(os.environ, S4G_RCCL_LIBRARY="/nonexistent/librccl.so.1"))
    assert r.returncode == 0, (r.returncode, r.stderr[-500:])
    rc, msg = r.stdout.strip().split(" ", 1)
    assert int(rc) == -1 and "RCCL is not available" in msg and "/nonexistent/librccl.so.1" in msg, r.stdout


def test_libswsharp_exports_only_its_c_api():
    """the host library exports the swsharp-compatible C API (version script libswsharp.map) and no C++ runtime instantiations
    (round 1 leaked std::thread state objects and vector bases into the dynamic symbol table)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "sift4g_amd", "lib", "libswsharp.so")],
                         capture_output=True, text=True, check=True).stdout
    names = [l.split()[-1] for l in out.splitlines() if l.strip()]
    assert names and not [n for n in names if n.startswith("_Z")]
    for must in ("alignDatabase", "chainCreate", "scorerCreateMatrix", "createEValueParams", "eValues", "dbAlignmentsMerge",
                 "shotgunDatabaseEValues", "readFastaChainsPart", "threadPoolSubmit", "cudaCheckCards"):
        assert must in names, must


def test_warmup_without_a_device_reports_it():
    """s4g_warmup is asynchronous, but the absence of a device is known at once (no thread is started)."""
    from sift4g_amd import _lib
    import torch
    if torch.cuda.device_count() == 0:
        assert _lib.lib.s4g_warmup(0) == _lib.S4G_ERR_NO_DEVICE
    assert _lib.lib.s4g_warmup(-1) == _lib.S4G_ERR_INVALID and _lib.lib.s4g_warmup(64) == _lib.S4G_ERR_INVALID
