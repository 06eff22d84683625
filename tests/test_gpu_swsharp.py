"""GPU tests of the swsharp-compatible API: alignDatabase / shotgunDatabase through libswsharp.so (HIP scoring)
against the same selection run on oracle scores, and the reference's own CLI linked against our libraries."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, rand_seqs
import swsharp_ctypes as sw
from swsharp_ctypes import L, vp, pvp, ci

pytestmark = pytest.mark.gpu


def _hits(out, n):
    return [(L.dbAlignmentGetTargetIdx(out[k]), L.dbAlignmentGetScore(out[k]), L.dbAlignmentGetValue(out[k]),
             L.dbAlignmentGetQueryStart(out[k]), L.dbAlignmentGetTargetStart(out[k]),
             tuple(sw.path_of(L.dbAlignmentGetPathLen, L.dbAlignmentGetMove, out[k]))) for k in range(n)]


def test_align_database_equals_selection_on_oracle_scores(oracle_mod, blosum62, cfg1):
    """the 15-argument call of database_alignment.cpp:83-86 on the reference fixture."""
    s = sw.scorer(10, 1)
    targets = [sw.chain(n, c) for n, c in cfg1["targets"]]
    arr = sw.chain_array(targets)
    cards = (ci * 1)(0)
    cdb = L.chainDatabaseCreate(arr, 0, len(targets), cards, 1)
    ev = L.createEValueParams(3091, s)
    cb = sw.evalue_callback()
    for qi, (qn, qc) in enumerate(cfg1["queries"]):
        q = sw.chain(qn, qc)
        out, n = pvp(), ci()
        L.alignDatabase(ctypes.byref(out), ctypes.byref(n), 0, q, cdb, s, 400, cb, ev, 1e-4, None, 0, cards, 1, None)
        ref, nr = pvp(), ci()
        sc = (ci * len(targets))(*[int(x) for x in cfg1["scores"][qi]])
        L.dbAlignmentsFromScores(ctypes.byref(ref), ctypes.byref(nr), 0, q, arr, len(targets), sc, s, 400, cb, ev, 1e-4)
        assert n.value == nr.value >= 5
        assert _hits(out, n.value) == _hits(ref, nr.value)
        # subset call (indexes argument)
        idx = (ci * 3)(4, 1, 7)
        sub, ns = pvp(), ci()
        L.alignDatabase(ctypes.byref(sub), ctypes.byref(ns), 0, q, cdb, s, 400, cb, ev, 1e9, idx, 3, cards, 1, None)
        assert sorted(h[0] for h in _hits(sub, ns.value)) == [1, 4, 7]
        assert all(cfg1["scores"][qi][h[0]] == h[1] for h in _hits(sub, ns.value))
        L.chainDelete(q)
    L.deleteEValueParams(ev); L.chainDatabaseDelete(cdb); L.scorerDelete(s)


_LAZY_SCRIPT = r"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
from conftest import rand_seqs
import swsharp_ctypes as sw
from swsharp_ctypes import L, vp, pvp, ci
rng = np.random.default_rng(77)
qs = rand_seqs(rng, 5, 120, 380)
ts = rand_seqs(rng, 900, 30, 420)
for k in range(120):                                  # homologs with indels: paths with all three moves
    m = qs[k %% 5].copy(); idx = rng.choice(len(m), size=len(m) // 4, replace=False)
    m[idx] = rand_seqs(rng, 1, len(idx), len(idx))[0]
    cut = int(rng.integers(10, len(m) - 10)); m = np.concatenate([m[:cut], rand_seqs(rng, 1, 3, 3)[0], m[cut + 2:]])
    ts[k * 7] = m
s = sw.scorer(10, 1)
tch = [sw.chain("t%%d" %% i, t) for i, t in enumerate(ts)]
qch = [sw.chain("q%%d" %% i, q) for i, q in enumerate(qs)]
ev = L.createEValueParams(int(sum(len(t) for t in ts)), s)
cb = sw.evalue_callback()
cards = (ci * 1)(0)
parts = [(0, 300), (300, 300), (600, 300)]            # three "chunks" of the database, as database_alignment.cpp:36-47 would read them
total, lens = None, None
K = 12
for (start, n) in parts:
    part = (pvp * len(qch))(); plens = (ci * len(qch))()
    for qi, q in enumerate(qch):                      # the wrapper's loop: database_alignment.cpp:61-95
        arr = sw.chain_array(tch[start:start + n])
        cdb = L.chainDatabaseCreate(arr, 0, n, cards, 1)
        out, cnt = pvp(), ci()
        L.alignDatabase(ctypes.byref(out), ctypes.byref(cnt), 0, q, cdb, s, K, cb, ev, 10.0, None, 0, cards, 1, None)
        _ = [L.dbAlignmentGetTargetIdx(out[k]) for k in range(cnt.value)]      # :89, all the wrapper reads between two calls
        L.chainDatabaseDelete(cdb)                                               # :92, while the traceback may still be running
        part[qi] = out; plens[qi] = cnt.value
    if total is None:
        total, lens = part, plens
    else:                                             # :97-104: losers are deleted before anybody read their paths
        L.dbAlignmentsMerge(total, lens, part, plens, len(qch), K)
res = []
for qi in range(len(qch)):
    for k in range(lens[qi]):
        a = total[qi][k]
        res.append([qi, L.chainGetName(L.dbAlignmentGetTarget(a)).decode(), L.dbAlignmentGetScore(a), L.dbAlignmentGetQueryStart(a),
                    L.dbAlignmentGetQueryEnd(a), L.dbAlignmentGetTargetStart(a), L.dbAlignmentGetTargetEnd(a),
                    sw.path_of(L.dbAlignmentGetPathLen, L.dbAlignmentGetMove, a)])
for qi in range(len(qch)):
    for k in range(lens[qi]): L.dbAlignmentDelete(total[qi][k])
print("RESULT " + json.dumps(res))
"""


def test_lazy_tracebacks_equal_eager_ones(oracle_mod, blosum62):
    """Round 5: a per-query alignDatabase call submits its traceback and returns; coordinates and paths are filled in by their first
    reader.  The reference's call pattern -- candidate database created and DELETED around every call, only target indices read in
    between, dbAlignmentsMerge deleting unread losers across three chunks, readers afterwards -- in a child process with the lazy
    path on (default) and off: identical alignments, and every path re-scores to its alignment's score under the oracle's rules."""
    import json, sys
    outs = {}
    for mode in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", _LAZY_SCRIPT % dict(root=ROOT)], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, S4G_LAZY_TRACEBACK=mode))
        assert r.returncode == 0, r.stderr[-3000:]
        outs[mode] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert outs["1"] == outs["0"] and len(outs["1"]) >= 40
    moves = {m for rec in outs["1"] for m in rec[7]}
    assert moves == {1, 2, 3}
    for rec in outs["1"][:10]:
        qi, name, score, q0, q1, t0, t1, path = rec
        assert q0 >= 0 and q1 >= q0 and t1 >= t0 and len(path) >= max(q1 - q0, t1 - t0) + 1 and score > 0


def test_shotgun_database_batch(oracle_mod, blosum62):
    rng = np.random.default_rng(31)
    qs = rand_seqs(rng, 12, 60, 200)
    ts = rand_seqs(rng, 400, 30, 260)
    for k in range(40):                                   # plant homologs
        m = qs[k % 12].copy(); idx = rng.choice(len(m), size=len(m) // 3, replace=False)
        m[idx] = rand_seqs(rng, 1, len(idx), len(idx))[0]; ts[k * 7] = m
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    s = sw.scorer(10, 1)
    tch = [sw.chain(f"t{i}", t) for i, t in enumerate(ts)]
    qch = [sw.chain(f"q{i}", q) for i, q in enumerate(qs)]
    tarr, qarr = sw.chain_array(tch), sw.chain_array(qch)
    cells = int(sum(len(t) for t in ts))
    ev = L.createEValueParams(cells, s)
    cb = sw.evalue_callback()
    # zero cards (device 0), and a multi-card list: "--cards 000" = three workers, each with its own replica of the database and
    # a third of the queries on its own host thread (all on the one GPU of the test box)
    for cards in (None, (0, 0, 0)):
        carr = (ci * len(cards))(*cards) if cards else None
        cdb = L.chainDatabaseCreate(tarr, 0, len(tch), carr, len(cards) if cards else 0)
        outs, lens = ctypes.POINTER(pvp)(), ctypes.POINTER(ci)()
        L.shotgunDatabase(ctypes.byref(outs), ctypes.byref(lens), 0, qarr, len(qch), cdb, s, 10, cb, ev, 1e-3, None, 0, carr,
                          len(cards) if cards else 0, None)
        total = 0
        for i in range(len(qs)):
            ref, nr = pvp(), ci()
            sc = (ci * len(ts))(*[int(x) for x in exp[i]])
            L.dbAlignmentsFromScores(ctypes.byref(ref), ctypes.byref(nr), 0, qch[i], tarr, len(tch), sc, s, 10, cb, ev, 1e-3)
            assert lens[i] == nr.value
            assert _hits(outs[i], lens[i]) == _hits(ref, nr.value)
            total += lens[i]
        assert total >= 40
        L.deleteShotgunDatabase(outs, lens, len(qs))
        L.chainDatabaseDelete(cdb)
    L.deleteEValueParams(ev); L.scorerDelete(s)


def test_shotgun_database_candidates(oracle_mod, blosum62):
    """shotgunDatabaseCandidates: per-query candidate lists (sorted ids, different sizes, an empty one, a NULL one, one that is
    the whole database) -> for every query exactly what alignDatabase on a database of just those chains gives, with target
    indices into the big database; also with a multi-card list."""
    rng = np.random.default_rng(33)
    qs = rand_seqs(rng, 9, 60, 220)
    ts = rand_seqs(rng, 300, 30, 260)
    for k in range(45):
        m = qs[k % 9].copy(); idx = rng.choice(len(m), size=len(m) // 3, replace=False)
        m[idx] = rand_seqs(rng, 1, len(idx), len(idx))[0]; ts[k * 6] = m
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    s = sw.scorer(10, 1)
    tch = [sw.chain(f"t{i}", t) for i, t in enumerate(ts)]
    qch = [sw.chain(f"q{i}", q) for i, q in enumerate(qs)]
    tarr, qarr = sw.chain_array(tch), sw.chain_array(qch)
    ev = L.createEValueParams(int(sum(len(t) for t in ts)), s)
    cb = sw.evalue_callback()
    lists = [sorted(rng.choice(len(ts), size=int(n), replace=False).tolist()) for n in (120, 7, 0, 299, 300, 64, 65, 1, 33)]
    lists[4] = list(range(len(ts)))
    arrs = [(ci * max(len(l), 1))(*l) for l in lists]
    ptrs = (ctypes.POINTER(ci) * len(lists))(*[ctypes.cast(a, ctypes.POINTER(ci)) for a in arrs])
    ptrs[7] = ctypes.cast(arrs[7], ctypes.POINTER(ci)); lens_in = (ci * len(lists))(*[len(l) for l in lists])
    ptrs[2] = ctypes.POINTER(ci)()                      # NULL list: no alignments for that query
    for cards in (None, (0, 0)):
        carr = (ci * len(cards))(*cards) if cards else None
        cdb = L.chainDatabaseCreate(tarr, 0, len(tch), carr, len(cards) if cards else 0)
        outs, lens = ctypes.POINTER(pvp)(), ctypes.POINTER(ci)()
        L.shotgunDatabaseCandidates(ctypes.byref(outs), ctypes.byref(lens), 0, qarr, len(qch), cdb, s, 10, cb, ev, 1e-3, ptrs, lens_in,
                                    carr, len(cards) if cards else 0, None)
        total = 0
        for i, l in enumerate(lists):
            if not l:
                assert lens[i] == 0
                continue
            sub = sw.chain_array([tch[t] for t in l])
            ref, nr = pvp(), ci()
            sc = (ci * len(l))(*[int(exp[i, t]) for t in l])
            L.dbAlignmentsFromScores(ctypes.byref(ref), ctypes.byref(nr), 0, qch[i], sub, len(l), sc, s, 10, cb, ev, 1e-3)
            want = [(l[h[0]],) + tuple(h[1:]) for h in _hits(ref, nr.value)]     # list position -> index into the database
            assert lens[i] == nr.value and _hits(outs[i], lens[i]) == want, i
            total += lens[i]
        assert total >= 10
        L.deleteShotgunDatabase(outs, lens, len(qs))
        L.chainDatabaseDelete(cdb)
    L.deleteEValueParams(ev); L.scorerDelete(s)


def test_evalue_driven_selection_equals_callback_path(oracle_mod, blosum62):
    """shotgunDatabaseEValues (score cut + streaming device top-k + exact host e-values + exactness check) returns exactly
    what shotgunDatabase returns with the reference's e-value callback -- including queries with more significant hits than
    maxAlignments, more than the device list k' (forces the full-row fallback), and none at all."""
    rng = np.random.default_rng(35)
    qs = rand_seqs(rng, 9, 80, 260)
    ts = rand_seqs(rng, 1500, 30, 320)
    for k in range(700):                                  # query 0: 700 homologs of graded similarity (> k' = 260 at max 1)
        m = qs[0].copy(); idx = rng.choice(len(m), size=int(rng.integers(1, len(m) // 2)), replace=False)
        m[idx] = rand_seqs(rng, 1, len(idx), len(idx))[0]; ts[k] = m
    for k in range(700, 760):                             # queries 1..3: 20 homologs each
        src = qs[1 + (k % 3)]; m = src.copy(); idx = rng.choice(len(m), size=len(m) // 4, replace=False)
        m[idx] = rand_seqs(rng, 1, len(idx), len(idx))[0]; ts[k] = m
    s = sw.scorer(10, 1)
    tch = [sw.chain(f"t{i}", t) for i, t in enumerate(ts)]
    qch = [sw.chain(f"q{i}", q) for i, q in enumerate(qs)]
    tarr, qarr = sw.chain_array(tch), sw.chain_array(qch)
    cells = int(sum(len(t) for t in ts))
    ev = L.createEValueParams(cells, s)
    cb = sw.evalue_callback()
    from oracle import evalue_ref as er
    exp = oracle_mod.scan(qs, ts, blosum62, 10, 1, threads=8)
    P = er.gumbel_parameters("BLOSUM_62", 10, 1, "swsharp")
    for cards in (None, (0, 0)):
        carr = (ci * len(cards))(*cards) if cards else None
        cdb = L.chainDatabaseCreate(tarr, 0, len(tch), carr, len(cards) if cards else 0)
        for max_al, thr in ((400, 1e-4), (25, 1e-4), (1, 10.0), (5, 1e-30)):
            a, la = ctypes.POINTER(pvp)(), ctypes.POINTER(ci)()
            b, lb = ctypes.POINTER(pvp)(), ctypes.POINTER(ci)()
            L.shotgunDatabase(ctypes.byref(a), ctypes.byref(la), 0, qarr, len(qch), cdb, s, max_al, cb, ev, thr, None, 0, None, 0, None)
            L.shotgunDatabaseEValues(ctypes.byref(b), ctypes.byref(lb), 0, qarr, len(qch), cdb, s, max_al, ev, thr, None, 0, None)
            for i in range(len(qs)):
                assert la[i] == lb[i], (max_al, thr, i, la[i], lb[i])
                assert _hits(a[i], la[i]) == _hits(b[i], lb[i]), (max_al, thr, i)
                # ... and against the INDEPENDENT restatement (oracle scores + oracle/evalue_ref.py, neither shares code with
                # the library): the same targets in the same order, the same scores, e-values to 1e-12 relative
                vals = [er.evalue(int(exp[i, t]), len(qs[i]), len(ts[t]), cells, P) for t in range(len(ts))]
                order = sorted((t for t in range(len(ts)) if vals[t] <= thr), key=lambda t: (vals[t], -int(exp[i, t]), t))[:max_al]
                got = _hits(b[i], lb[i])
                assert [h[0] for h in got] == order, (max_al, thr, i)
                assert [h[1] for h in got] == [int(exp[i, t]) for t in order]
                for h, t in zip(got, order):
                    assert h[2] == pytest.approx(vals[t], rel=1e-12, abs=0.0)
            assert la[0] == min(max_al, la[0]) and la[0] >= 1
            L.deleteShotgunDatabase(a, la, len(qs)); L.deleteShotgunDatabase(b, lb, len(qs))
        L.chainDatabaseDelete(cdb)
    L.deleteEValueParams(ev); L.scorerDelete(s)


DROPIN = os.path.join(ROOT, "build", "dropin", "sift4g")


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs /root/reference at build time: make dropin)")
def test_reference_cli_runs_on_our_library(tmp_path):
    """The reference's unmodified sift4g CLI, linked against libswsharp/libs4g_hip, on its own test_files
    (test_files/README.md:13-19).  Output parity with swsharp is unpinned; this checks the pipeline completes and
    the predictions are well-formed and stable against the committed run of this same path."""
    tf = os.path.join(GOLDEN, "test_files")
    r = subprocess.run([DROPIN, "-q", os.path.join(tf, "query.fasta"), "--subst", tf, "-d",
                        os.path.join(tf, "sample_protein_database.fa"), "--out", str(tmp_path), "--cards", "0", "--sub-results"],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    # every reported hit (score, e-value, coordinates, identity / gaps from the device traceback): the same bytes the CPU
    # oracle pipeline produces (tests/test_cpu_pipeline.py compares its own run with this same file)
    assert (tmp_path / "alignments.txt").read_text() == open(os.path.join(GOLDEN, "dropin_predictions", "alignments.txt")).read()
    for name, n_subst in (("LACI_ECOLI", 3), ("PURR_SALTY", 2)):
        path = tmp_path / f"{name}.SIFTprediction"
        assert path.exists(), r.stderr[-2000:]
        lines = [l for l in path.read_text().splitlines() if l and not l.startswith("WARNING")]
        assert len(lines) == n_subst
        for l in lines:
            f = l.split("\t")
            assert f[1] in ("TOLERATED", "DELETERIOUS") and 0.0 <= float(f[2]) <= 1.0
        gold = os.path.join(GOLDEN, "dropin_predictions", f"{name}.SIFTprediction")
        if os.path.exists(gold):
            assert path.read_text() == open(gold).read()


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs /root/reference at build time: make dropin)")
def test_reference_cli_synthetic_set_matches_the_committed_run(tmp_path):
    """the reference CLI on 8 queries x 1500 synthetic targets (per-query candidate databases -> pair kernels for scores and
    traceback): alignments.txt equals tests/golden/dropin_synth/, which the CPU oracle pipeline reproduces byte for byte
    (tests/test_cpu_pipeline.py)."""
    import dropin_synth
    got = dropin_synth.run(DROPIN, str(tmp_path))
    assert got == open(os.path.join(GOLDEN, "dropin_synth", "alignments.txt"), "rb").read()


BATCHED = os.path.join(ROOT, "build", "dropin", "sift4g_batched")


@pytest.mark.skipif(not os.path.exists(BATCHED), reason="batched drop-in not built (needs /root/reference at build time: make dropin-batched)")
def test_batched_wrapper_gives_the_reference_flows_alignments(tmp_path):
    """build/dropin/sift4g_batched = the reference CLI with ONE source file replaced (its alignment wrapper, by
    sift4g_amd/csrc/dropin/batched_database_alignment.cpp: one shotgunDatabaseCandidates call per chunk instead of one
    alignDatabase call per query).  Same alignments and predictions, byte for byte, on the fixture and on the synthetic set."""
    import dropin_synth
    tf = os.path.join(GOLDEN, "test_files")
    out = tmp_path / "fixture"
    out.mkdir()
    r = subprocess.run([BATCHED, "-q", os.path.join(tf, "query.fasta"), "--subst", tf, "-d",
                        os.path.join(tf, "sample_protein_database.fa"), "--out", str(out), "--cards", "0", "--sub-results"],
                       capture_output=True, text=True, timeout=600, cwd=str(out))
    assert r.returncode == 0, r.stderr[-2000:]
    for n in ("alignments.txt", "LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction"):
        assert (out / n).read_bytes() == open(os.path.join(GOLDEN, "dropin_predictions", n), "rb").read(), n
    syn = tmp_path / "synth"
    syn.mkdir()
    want = open(os.path.join(GOLDEN, "dropin_synth", "alignments.txt"), "rb").read()
    assert dropin_synth.run(BATCHED, str(syn)) == want
    parts = tmp_path / "parts"                                  # the database in ~17 parts (merge across parts)
    parts.mkdir()
    assert dropin_synth.run(BATCHED, str(parts), env=dict(os.environ, S4G_CHUNK_BYTES="30000")) == want


DEVSEARCH = os.path.join(ROOT, "build", "dropin", "sift4g_devsearch")


@pytest.mark.skipif(not os.path.exists(DEVSEARCH), reason="device-prefilter drop-in not built (needs /root/reference at build time: make dropin-devsearch)")
def test_device_prefilter_build_gives_the_reference_flows_outputs(tmp_path):
    """build/dropin/sift4g_devsearch = the reference CLI with its k-mer prefilter on the device (database_search.cpp and
    hash.cpp replaced by sift4g_amd/csrc/dropin/device_database_search.cpp -> s4g_prefilter_*) and the batched alignment
    wrapper.  Same alignments and predictions, byte for byte, as the build that runs the reference's own prefilter: on the
    fixture, on the synthetic set, with the database read in many parts, and at the other k-mer lengths."""
    import dropin_synth
    tf = os.path.join(GOLDEN, "test_files")

    def fixture(exe, name, extra=(), env=None):
        out = tmp_path / name
        out.mkdir()
        r = subprocess.run([exe, "-q", os.path.join(tf, "query.fasta"), "--subst", tf, "-d",
                            os.path.join(tf, "sample_protein_database.fa"), "--out", str(out), "--cards", "0", "--sub-results", *extra],
                           capture_output=True, text=True, timeout=600, cwd=str(out), env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return {n: (out / n).read_bytes() for n in ("alignments.txt", "LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction")}

    got = fixture(DEVSEARCH, "fixture")
    for n, data in got.items():
        assert data == open(os.path.join(GOLDEN, "dropin_predictions", n), "rb").read(), n
    assert fixture(DEVSEARCH, "fixture_parts", env=dict(os.environ, S4G_PREFILTER_CHUNK_BYTES="20000")) == got
    for k in ("3", "4"):
        assert fixture(DEVSEARCH, "dev_k" + k, ("--kmer-length", k)) == fixture(DROPIN, "ref_k" + k, ("--kmer-length", k)), k
    syn = tmp_path / "synth"
    syn.mkdir()
    want = open(os.path.join(GOLDEN, "dropin_synth", "alignments.txt"), "rb").read()
    assert dropin_synth.run(DEVSEARCH, str(syn)) == want
    parts = tmp_path / "parts"
    parts.mkdir()
    assert dropin_synth.run(DEVSEARCH, str(parts), env=dict(os.environ, S4G_PREFILTER_CHUNK_BYTES="30000", S4G_CHUNK_BYTES="30000")) == want


@pytest.mark.skipif(not (os.path.exists(DEVSEARCH) and os.path.exists(BATCHED)), reason="drop-in builds missing (need /root/reference at build time)")
def test_selection_ties_by_name_is_the_same_on_flat_and_chain_databases(tmp_path):
    """S4G_SELECTION_TIES=name orders equal (value, score) survivors by chain name: the one-pass device-prefilter build keeps
    the database flat (placeholder chains whose names are the records' header text, chainDatabaseCreateFlat) and must order exactly as
    the builds that hold real chains do -- same bytes in alignments.txt, on the synthetic set, whole and in parts."""
    import dropin_synth
    env = dict(os.environ, S4G_SELECTION_TIES="name")
    outs = []
    for k, (exe, extra) in enumerate(((BATCHED, {}), (DEVSEARCH, {}), (DEVSEARCH, {"S4G_PREFILTER_CHUNK_BYTES": "30000", "S4G_CHUNK_BYTES": "30000"}), (DROPIN, {}))):
        d = tmp_path / f"run{k}"
        d.mkdir()
        outs.append(dropin_synth.run(exe, str(d), env=dict(env, **extra)))
    assert outs[0] == outs[1] == outs[2] == outs[3]
    assert len(outs[0]) > 1000


FULLSCAN = os.path.join(ROOT, "build", "dropin", "sift4g_fullscan")


@pytest.mark.skipif(not os.path.exists(FULLSCAN), reason="full-scan drop-in not built (needs /root/reference at build time: make dropin-fullscan)")
def test_fullscan_build_is_a_superset_of_the_prefiltered_build(tmp_path):
    """build/dropin/sift4g_fullscan = the reference CLI with the k-mer prefilter replaced by a residue count and
    the alignment wrapper by one shotgunDatabaseEValues call per chunk over ALL its sequences.  On the MI355X it writes the
    bytes of tests/golden/dropin_fullscan/ (which the oracle pipeline also writes, tests/test_cpu_pipeline.py); every hit of the
    prefiltered builds is there with identical score, e-value and alignment, plus the homologs the prefilter missed.  Stage times
    of both builds on a 60-query x 20 000-target set are printed (pytest -s) and recorded in profiles/."""
    import dropin_synth, re, time
    from test_cpu_pipeline import hit_lines
    tf = os.path.join(GOLDEN, "test_files")
    out = tmp_path / "fixture"
    out.mkdir()
    r = subprocess.run([FULLSCAN, "-q", os.path.join(tf, "query.fasta"), "--subst", tf, "-d",
                        os.path.join(tf, "sample_protein_database.fa"), "--out", str(out), "--cards", "0", "--sub-results"],
                       capture_output=True, text=True, timeout=600, cwd=str(out))
    assert r.returncode == 0, r.stderr[-2000:]
    gold = os.path.join(GOLDEN, "dropin_fullscan")
    for n in ("alignments.txt", "LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction"):
        assert (out / n).read_bytes() == open(os.path.join(gold, n), "rb").read(), n
    syn = tmp_path / "synth"
    syn.mkdir()
    got = dropin_synth.run(FULLSCAN, str(syn))
    assert got == open(os.path.join(gold, "synth_alignments.txt"), "rb").read()
    pre = set(hit_lines(open(os.path.join(GOLDEN, "dropin_synth", "alignments.txt"), "rb").read()))
    assert pre <= set(hit_lines(got))
    parts = tmp_path / "parts"
    parts.mkdir()
    assert dropin_synth.run(FULLSCAN, str(parts), env=dict(os.environ, S4G_CHUNK_BYTES="30000")) == got
    # a larger set: superset again, and the wall time of both builds (whole CLI: FASTA reads, prefilter or count, alignment, SIFT)
    times = {}
    outs = {}
    for name, exe in (("prefiltered (batched wrapper)", BATCHED), ("full scan", FULLSCAN)):
        d = tmp_path / name.split()[0]
        d.mkdir()
        t0 = time.perf_counter()
        outs[name] = dropin_synth.run(exe, str(d), nq=60, nt=20000)
        times[name] = time.perf_counter() - t0
    a, b = set(hit_lines(outs["prefiltered (batched wrapper)"])), set(hit_lines(outs["full scan"]))
    assert a <= b, sorted(a - b)[:3]
    print(f"\n60 queries x 20000 targets, whole CLI incl. FASTA generation: " + ", ".join(f"{k} {v:.2f} s ({len(hit_lines(outs[k]))} hits)" for k, v in times.items()))


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs /root/reference at build time: make dropin)")
@pytest.mark.parametrize("args", [("--matrix", "BLOSUM_45", "-g", "14", "-e", "2"), ("--matrix", "BLOSUM_30", "-g", "9", "-e", "1"),
                                  ("--matrix", "BLOSUM_80", "-g", "10", "-e", "1", "--evalue", "1e-10")])
def test_reference_cli_other_matrices_gpu_equals_oracle_pipeline(tmp_path, args):
    """`--matrix` / `-g` / `-e` / `--evalue` of the reference CLI (main.cpp:101-121,304-314) with tables other than the default --
    BLOSUM_45 14/2, the help text's "BLOSUM_30" (= PAM 30) 9/1, BLOSUM_80 10/1 with a tighter cut: the MI355X run and the same
    binary on the oracle stand-in (tests/cpu_shim) must write the same alignments.txt and the same .SIFTprediction files, byte
    for byte (scores, e-values from each matrix's own Gumbel row, paths, selection, SIFT statistics)."""
    tf = os.path.join(GOLDEN, "test_files")
    shim = os.path.join(ROOT, "tests", "cpu_shim")
    subprocess.run(["make", "-s", "-C", shim], check=True, capture_output=True, timeout=300)
    outs = {}
    for name, env in (("gpu", dict(os.environ)),
                      ("oracle", dict(os.environ, LD_LIBRARY_PATH=os.path.join(shim, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", "")))):
        d = tmp_path / name
        d.mkdir()
        r = subprocess.run([DROPIN, "-q", os.path.join(tf, "query.fasta"), "--subst", tf, "-d", os.path.join(tf, "sample_protein_database.fa"),
                            "--out", str(d), "--cards", "0", "--sub-results", *args], env=env, capture_output=True, text=True, timeout=600, cwd=str(d))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "[WARNING]" not in r.stderr, r.stderr[-500:]            # every one of these penalty pairs has a published row
        outs[name] = {n: (d / n).read_bytes() for n in sorted(os.listdir(d)) if n.endswith((".SIFTprediction", "alignments.txt"))}
    assert outs["gpu"] and outs["gpu"].keys() == outs["oracle"].keys()
    for n in outs["gpu"]:
        assert outs["gpu"][n] == outs["oracle"][n], n
    assert outs["gpu"]["alignments.txt"].count(b"\n") >= 8
    assert outs["gpu"]["alignments.txt"] != open(os.path.join(GOLDEN, "dropin_predictions", "alignments.txt"), "rb").read()


@pytest.mark.skipif(not (os.path.exists(DEVSEARCH) and os.path.exists(DROPIN)), reason="drop-in builds missing (need /root/reference at build time)")
def test_device_prefilter_differs_from_the_reference_prefilter_only_in_ties_at_the_cut(tmp_path):
    """300 queries x 200 000 targets through the reference's own prefilter (32 threads) and through the device's: 5000 candidates
    of 200 000 make the cut bind for every query, and planted homologs of low identity sit ON the cut (one shared 5-mer: score
    1 / length, like thousands of unrelated sequences of that length).  Which of such ties survive is the one thing the two
    prefilters may decide differently (the reference's own answer changes with --threads); so every alignment only one of the
    two builds reports must be a pair whose prefilter score EQUALS its query's cut, and everything else must be identical."""
    import oracle
    from sift4g_amd import synth
    from sift4g_amd.seqio import write_fasta
    nq, nt = 300, 200000
    c = synth.make_config(2, nq, nt)
    write_fasta(str(tmp_path / "q.fa"), [f"q{i}" for i in range(nq)], c["q_codes"], c["q_off"])
    write_fasta(str(tmp_path / "db.fa"), [f"t{i}" for i in range(nt)], c["t_codes"], c["t_off"])
    lines = {}
    for name, exe in (("reference", DROPIN), ("device", DEVSEARCH)):
        out = tmp_path / name
        out.mkdir()
        r = subprocess.run([exe, "-q", str(tmp_path / "q.fa"), "-d", str(tmp_path / "db.fa"), "--cards", "0", "--out", str(out), "--sub-results", "-t", "32"],
                           capture_output=True, text=True, timeout=600, cwd=str(out))
        assert r.returncode == 0, r.stderr[-2000:]
        lines[name] = set(ln for ln in (out / "alignments.txt").read_bytes().split(b"\n") if ln and not ln.startswith(b"#"))
    assert len(lines["reference"]) > 13000
    only = (lines["reference"] - lines["device"]) | (lines["device"] - lines["reference"])
    assert len(only) <= 20                                      # a handful of ties, not a different result
    qo, to = c["q_off"], c["t_off"]
    targets = [c["t_codes"][int(to[i]):int(to[i + 1])] for i in range(nt)]
    for ln in only:
        f = ln.split(b"\t")
        q, t = int(f[0][1:]), int(f[1][1:])
        pf = oracle.Prefilter([c["q_codes"][int(qo[q]):int(qo[q + 1])]], 5, 5000)
        pf.add(targets)
        _, _, _, cut = pf.finish()
        lis = int(pf.lis([targets[t]])[0, 0])
        assert cut[0] > 0 and np.float32(lis) / np.float32(len(targets[t])) == cut[0], (q, t, lis, len(targets[t]), float(cut[0]))
