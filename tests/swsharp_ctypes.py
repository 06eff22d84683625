"""ctypes binding of sift4g_amd/lib/libswsharp.so (include/swsharp/*.h) for the tests: plays the role of a
C/C++ host program written against the swsharp API (as sift4g is)."""
import ctypes
import os

from conftest import ROOT

import sift4g_amd._lib  # noqa: F401  loads libs4g_hip.so first (dependency of libswsharp.so)

PATH = os.path.join(ROOT, "sift4g_amd", "lib", "libswsharp.so")
L = ctypes.CDLL(PATH)

vp, ci, cd, cc = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_char
pvp, pi = ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)
ValueFunction = ctypes.CFUNCTYPE(None, ctypes.POINTER(cd), pi, vp, pvp, ci, pi, ci, vp)
Routine = ctypes.CFUNCTYPE(vp, vp)

SIG = {
    "chainCreate": (vp, [ctypes.c_char_p, ci, ctypes.c_char_p, ci]),
    "chainCreateView": (vp, [vp, ci, ci, ci]),
    "chainDelete": (None, [vp]),
    "chainGetLength": (ci, [vp]),
    "chainGetChar": (cc, [vp, ci]),
    "chainGetCode": (cc, [vp, ci]),
    "chainGetName": (ctypes.c_char_p, [vp]),
    "chainGetCodes": (ctypes.POINTER(ctypes.c_ubyte), [vp]),
    "readFastaChains": (None, [ctypes.POINTER(pvp), pi, ctypes.c_char_p]),
    "readFastaChainsPartInit": (None, [ctypes.POINTER(pvp), pi, pvp, pi, ctypes.c_char_p]),
    "readFastaChainsPart": (ci, [ctypes.POINTER(pvp), pi, vp, ci, ctypes.c_size_t]),
    "deleteFastaChains": (None, [pvp, ci]),
    "readFastaCodesPart": (ci, [ctypes.POINTER(ctypes.POINTER(ctypes.c_ubyte)), ctypes.POINTER(ctypes.POINTER(ctypes.c_size_t)), pi, vp, ci, ctypes.c_size_t]),
    "fastaResidueCount": (None, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]),
    "scorerCreateMatrix": (None, [pvp, ctypes.c_char_p, ci, ci]),
    "scorerDelete": (None, [vp]),
    "scorerGetGapOpen": (ci, [vp]),
    "scorerGetGapExtend": (ci, [vp]),
    "scorerScore": (ci, [vp, cc, cc]),
    "scorerGetMaxScore": (ci, [vp]),
    "cudaGetCards": (None, [ctypes.POINTER(pi), pi]),
    "cudaCheckCards": (ci, [pi, ci]),
    "swsharpPreferredCard": (ci, []),
    "chainDatabaseCreate": (vp, [pvp, ci, ci, pi, ci]),
    "chainDatabaseDelete": (None, [vp]),
    "alignDatabase": (None, [ctypes.POINTER(pvp), pi, ci, vp, vp, vp, ci, ValueFunction, vp, cd, pi, ci, pi, ci, vp]),
    "shotgunDatabase": (None, [ctypes.POINTER(ctypes.POINTER(pvp)), ctypes.POINTER(pi), ci, pvp, ci, vp, vp, ci,
                               ValueFunction, vp, cd, pi, ci, pi, ci, vp]),
    "shotgunDatabaseCandidates": (None, [ctypes.POINTER(ctypes.POINTER(pvp)), ctypes.POINTER(pi), ci, pvp, ci, vp, vp, ci,
                                         ValueFunction, vp, cd, ctypes.POINTER(pi), pi, pi, ci, vp]),
    "shotgunDatabaseEValues": (None, [ctypes.POINTER(ctypes.POINTER(pvp)), ctypes.POINTER(pi), ci, pvp, ci, vp, vp, ci, vp, cd, pi, ci, vp]),
    "dbAlignmentsFromScores": (None, [ctypes.POINTER(pvp), pi, ci, vp, pvp, ci, pi, vp, ci, ValueFunction, vp, cd]),
    "alignPair": (None, [pvp, ci, vp, vp, vp, pi, ci, vp]),
    "dbAlignmentGetQueryStart": (ci, [vp]), "dbAlignmentGetQueryEnd": (ci, [vp]),
    "dbAlignmentGetTarget": (vp, [vp]), "dbAlignmentGetTargetIdx": (ci, [vp]),
    "dbAlignmentGetTargetStart": (ci, [vp]), "dbAlignmentGetTargetEnd": (ci, [vp]),
    "dbAlignmentGetScore": (ci, [vp]), "dbAlignmentGetValue": (cd, [vp]),
    "dbAlignmentGetPathLen": (ci, [vp]), "dbAlignmentGetMove": (cc, [vp, ci]),
    "dbAlignmentToAlignment": (vp, [vp]), "dbAlignmentDelete": (None, [vp]), "dbAlignmentGetScorer": (vp, [vp]),
    "dbAlignmentsMerge": (None, [ctypes.POINTER(pvp), pi, ctypes.POINTER(pvp), pi, ci, ci]),
    "deleteShotgunDatabase": (None, [ctypes.POINTER(pvp), pi, ci]),
    "outputShotgunDatabase": (None, [ctypes.POINTER(pvp), pi, ci, ctypes.c_char_p, ci]),
    "alignmentGetQueryStart": (ci, [vp]), "alignmentGetQueryEnd": (ci, [vp]),
    "alignmentGetTargetStart": (ci, [vp]), "alignmentGetTargetEnd": (ci, [vp]),
    "alignmentGetScore": (ci, [vp]), "alignmentGetPathLen": (ci, [vp]), "alignmentGetMove": (cc, [vp, ci]),
    "alignmentGetQuery": (vp, [vp]), "alignmentGetTarget": (vp, [vp]),
    "alignmentDelete": (None, [vp]),
    "threadPoolInitialize": (ci, [ci]), "threadPoolTerminate": (None, []),
    "threadPoolSubmit": (vp, [Routine, vp]), "threadPoolTaskWait": (None, [vp]), "threadPoolTaskDelete": (None, [vp]),
    "createEValueParams": (vp, [ctypes.c_longlong, vp]), "deleteEValueParams": (None, [vp]),
    "eValues": (None, [ctypes.POINTER(cd), pi, vp, pvp, ci, pi, ci, vp]),
}
for name, (res, args) in SIG.items():
    fn = getattr(L, name)
    fn.restype, fn.argtypes = res, args

MOVE_DIAG, MOVE_LEFT, MOVE_UP = 1, 2, 3
ORACLE_TO_MOVE = {0: MOVE_DIAG, 1: MOVE_LEFT, 2: MOVE_UP}

libc = ctypes.CDLL(None)
libc.free.argtypes = [vp]
libc.malloc.restype = vp
libc.malloc.argtypes = [ctypes.c_size_t]


def chain(name, codes):
    s = bytes((int(c) + 65) for c in codes)
    return L.chainCreate(name.encode(), len(name), s, len(s))


def chain_codes(c):
    n = L.chainGetLength(c)
    p = L.chainGetCodes(c)
    return [p[i] for i in range(n)]


def scorer(go=10, ge=1, name="BLOSUM_62"):
    s = vp()
    L.scorerCreateMatrix(ctypes.byref(s), name.encode(), go, ge)
    return s


def chain_array(chains):
    arr = (vp * max(len(chains), 1))(*chains)
    return arr


def path_of(getlen, getmove, a):
    return [ord(getmove(a, i)) for i in range(getlen(a))]


def evalue_callback():
    """the reference's valueFunction (database_alignment.cpp:129-134): forwards to eValues with param = EValueParams*"""
    def fn(values, scores, query, database, n, cards, cards_len, param):
        L.eValues(values, scores, query, database, n, cards, cards_len, param)
    return ValueFunction(fn)
