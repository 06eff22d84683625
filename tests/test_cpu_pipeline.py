"""End-to-end equality of the two pipelines on the reference's own fixture, without a GPU: the reference's unmodified CLI
(build/dropin/sift4g, compiled in place from /root/reference by `make dropin`) is run with a TEST-ONLY stand-in for
libs4g_hip.so that answers the device ABI with the CPU oracle (tests/cpu_shim/).  Its .SIFTprediction files must be byte
for byte the ones the same CLI wrote on an MI355X through the HIP kernels (tests/golden/dropin_predictions/) -- "GPU path ==
its own CPU path", the strongest end-to-end statement available while swsharp is absent (parity unpinned)."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT
import dropin_synth

DROPIN = os.path.join(ROOT, "build", "dropin", "sift4g")
FULLSCAN = os.path.join(ROOT, "build", "dropin", "sift4g_fullscan")
SHIM = os.path.join(ROOT, "tests", "cpu_shim")


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs /root/reference at build time: make dropin)")
def test_reference_cli_on_the_cpu_oracle_reproduces_the_gpu_predictions(tmp_path):
    subprocess.run(["make", "-s", "-C", SHIM], check=True, capture_output=True, timeout=300)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(SHIM, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    ldd = subprocess.run(["ldd", DROPIN], env=env, capture_output=True, text=True).stdout
    assert os.path.join(SHIM, "lib", "libs4g_hip.so") in ldd, ldd          # the stand-in, not the HIP library, is what gets loaded
    files = os.path.join(GOLDEN, "test_files")
    r = subprocess.run([DROPIN, "-q", os.path.join(files, "query.fasta"), "--subst", files + os.sep,
                        "-d", os.path.join(files, "sample_protein_database.fa"), "--cards", "0", "--out", str(tmp_path), "--sub-results"],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = sorted(f for f in os.listdir(tmp_path) if f.endswith(".SIFTprediction"))
    assert names == ["LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction"]
    for n in names + ["alignments.txt"]:     # alignments.txt: scores, e-values, coordinates, identity / gap counts of every hit (bm9)
        assert (tmp_path / n).read_bytes() == open(os.path.join(GOLDEN, "dropin_predictions", n), "rb").read(), n


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs /root/reference at build time: make dropin)")
def test_reference_cli_on_the_cpu_oracle_reproduces_the_gpu_alignments_of_a_synthetic_set(tmp_path):
    """8 queries x 1500 synthetic targets with planted, gapped homologs: 250 reported alignments, byte-identical between the
    oracle pipeline (here) and the MI355X run (tests/golden/dropin_synth/, re-checked on the GPU by tests/test_gpu_swsharp.py)."""
    subprocess.run(["make", "-s", "-C", SHIM], check=True, capture_output=True, timeout=300)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(SHIM, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    got = dropin_synth.run(DROPIN, str(tmp_path), env=env)
    want = open(os.path.join(GOLDEN, "dropin_synth", "alignments.txt"), "rb").read()
    assert got == want
    assert got.count(b"\n") == 250


BATCHED = os.path.join(ROOT, "build", "dropin", "sift4g_batched")


@pytest.mark.skipif(not os.path.exists(BATCHED), reason="batched drop-in not built (needs /root/reference at build time: make dropin-batched)")
def test_batched_wrapper_on_the_cpu_oracle_reproduces_the_same_outputs(tmp_path):
    """the reference CLI with its alignment wrapper replaced by sift4g_amd/csrc/dropin/batched_database_alignment.cpp (one
    shotgunDatabaseCandidates call per chunk): identical files to the per-query flow, here on the oracle stand-in."""
    subprocess.run(["make", "-s", "-C", SHIM], check=True, capture_output=True, timeout=300)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(SHIM, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    files = os.path.join(GOLDEN, "test_files")
    out = tmp_path / "fixture"
    out.mkdir()
    r = subprocess.run([BATCHED, "-q", os.path.join(files, "query.fasta"), "--subst", files + os.sep,
                        "-d", os.path.join(files, "sample_protein_database.fa"), "--cards", "0", "--out", str(out), "--sub-results"],
                       env=env, cwd=str(out), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for n in ("alignments.txt", "LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction"):
        assert (out / n).read_bytes() == open(os.path.join(GOLDEN, "dropin_predictions", n), "rb").read(), n
    syn = tmp_path / "synth"
    syn.mkdir()
    want = open(os.path.join(GOLDEN, "dropin_synth", "alignments.txt"), "rb").read()
    assert dropin_synth.run(BATCHED, str(syn), env=env) == want
    # the database in ~17 parts: consumed candidate prefixes, dbAlignmentsMerge across parts, deletion of unused chains
    parts = tmp_path / "parts"
    parts.mkdir()
    assert dropin_synth.run(BATCHED, str(parts), env=dict(env, S4G_CHUNK_BYTES="30000")) == want


DEVSEARCH = os.path.join(ROOT, "build", "dropin", "sift4g_devsearch")


@pytest.mark.skipif(not os.path.exists(DEVSEARCH), reason="device-prefilter drop-in not built (needs /root/reference at build time: make dropin-devsearch)")
def test_device_prefilter_build_on_the_cpu_oracle_reproduces_the_same_outputs(tmp_path):
    """the reference CLI with its k-mer prefilter (database_search.cpp, hash.cpp) replaced by
    sift4g_amd/csrc/dropin/device_database_search.cpp and its alignment wrapper by the batched one: the same files as the
    reference's own prefilter gives (its candidates differ from the replacement's only in ties at the cut, and on these inputs
    no such tie decides a reported hit), here with the oracle stand-in answering s4g_prefilter_* (oracle/kmer_prefilter.c)."""
    subprocess.run(["make", "-s", "-C", SHIM], check=True, capture_output=True, timeout=300)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(SHIM, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    files = os.path.join(GOLDEN, "test_files")
    out = tmp_path / "fixture"
    out.mkdir()
    r = subprocess.run([DEVSEARCH, "-q", os.path.join(files, "query.fasta"), "--subst", files + os.sep,
                        "-d", os.path.join(files, "sample_protein_database.fa"), "--cards", "0", "--out", str(out), "--sub-results"],
                       env=env, cwd=str(out), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for n in ("alignments.txt", "LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction"):
        assert (out / n).read_bytes() == open(os.path.join(GOLDEN, "dropin_predictions", n), "rb").read(), n
    syn = tmp_path / "synth"
    syn.mkdir()
    want = open(os.path.join(GOLDEN, "dropin_synth", "alignments.txt"), "rb").read()
    assert dropin_synth.run(DEVSEARCH, str(syn), env=env) == want
    # the database read in ~17 parts by both stages: running candidate lists across parts, reading overlapped with the prefilter
    parts = tmp_path / "parts"
    parts.mkdir()
    assert dropin_synth.run(DEVSEARCH, str(parts), env=dict(env, S4G_PREFILTER_CHUNK_BYTES="30000", S4G_CHUNK_BYTES="30000")) == want
    # (the runs above align out of the flat parts the prefilter's pass left behind -- one pass over the file, chains only for the
    # reported targets; the same with the file read a second time into chains, and with a budget too small to keep the parts)
    for k, extra in enumerate((dict(S4G_READ_DATABASE_TWICE="1"), dict(S4G_FLAT_DB_BYTES="20000", S4G_PREFILTER_CHUNK_BYTES="30000"))):
        d = tmp_path / f"twice{k}"
        d.mkdir()
        assert dropin_synth.run(DEVSEARCH, str(d), env=dict(env, **extra)) == want


FULLSCAN = os.path.join(ROOT, "build", "dropin", "sift4g_fullscan")


def hit_lines(data):
    return [l for l in data.splitlines() if l and not l.startswith(b"#")]


@pytest.mark.skipif(not os.path.exists(FULLSCAN), reason="full-scan drop-in not built (needs /root/reference at build time: make dropin-fullscan)")
def test_fullscan_build_on_the_cpu_oracle(tmp_path):
    """the reference CLI with the k-mer prefilter and the alignment wrapper replaced (make dropin-fullscan: every query against
    every sequence of every chunk through shotgunDatabaseEValues), here on the oracle stand-in: reproduces
    tests/golden/dropin_fullscan/ (which the MI355X run must also reproduce), and every hit of the prefiltered builds is in it
    unchanged -- same score, e-value, coordinates, identity and gap counts -- next to the homologs the prefilter missed."""
    subprocess.run(["make", "-s", "-C", SHIM], check=True, capture_output=True, timeout=300)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(SHIM, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    files = os.path.join(GOLDEN, "test_files")
    out = tmp_path / "fixture"
    out.mkdir()
    r = subprocess.run([FULLSCAN, "-q", os.path.join(files, "query.fasta"), "--subst", files + os.sep,
                        "-d", os.path.join(files, "sample_protein_database.fa"), "--cards", "0", "--out", str(out), "--sub-results"],
                       env=env, cwd=str(out), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    gold = os.path.join(GOLDEN, "dropin_fullscan")
    for n in ("alignments.txt", "LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction"):
        assert (out / n).read_bytes() == open(os.path.join(gold, n), "rb").read(), n
    full = set(hit_lines((out / "alignments.txt").read_bytes()))
    pre = set(hit_lines(open(os.path.join(GOLDEN, "dropin_predictions", "alignments.txt"), "rb").read()))
    assert pre < full and len(full - pre) == 5
    syn = tmp_path / "synth"
    syn.mkdir()
    got = dropin_synth.run(FULLSCAN, str(syn), env=env)
    assert got == open(os.path.join(gold, "synth_alignments.txt"), "rb").read()
    pre = set(hit_lines(open(os.path.join(GOLDEN, "dropin_synth", "alignments.txt"), "rb").read()))
    assert pre <= set(hit_lines(got)) and len(hit_lines(got)) == len(pre) + 6
    parts = tmp_path / "parts"                     # the database in ~17 parts: dbAlignmentsMerge across parts, chain deletion
    parts.mkdir()
    assert dropin_synth.run(FULLSCAN, str(parts), env=dict(env, S4G_CHUNK_BYTES="30000")) == got


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs /root/reference at build time: make dropin)")
def test_reference_cli_with_the_serialized_database_cache(tmp_path):
    """S4G_FASTA_CACHE=1: the prefilter's pass over the database (database_search.cpp:81-96) leaves `<db>.s4gchains`, the alignment
    stage of the SAME run reads it (database_alignment.cpp:36-47, serialized = 1), and so does a second run: both write the files
    of the uncached run, byte for byte."""
    import shutil
    subprocess.run(["make", "-s", "-C", SHIM], check=True, capture_output=True, timeout=300)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(SHIM, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""), S4G_FASTA_CACHE="1")
    files = os.path.join(GOLDEN, "test_files")
    db = tmp_path / "db.fa"
    shutil.copy(os.path.join(files, "sample_protein_database.fa"), db)
    qdir = tmp_path / "q"                          # the query file too is read through the caching readers: keep its cache out of the fixtures
    qdir.mkdir()
    query = str(qdir / "query.fasta")
    shutil.copy(os.path.join(files, "query.fasta"), query)
    for run in ("first", "second"):
        out = tmp_path / run
        out.mkdir()
        r = subprocess.run([DROPIN, "-q", query, "--subst", files + os.sep, "-d", str(db), "--cards", "0",
                            "--out", str(out), "--sub-results"], env=env, cwd=str(out), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert (tmp_path / "db.fa.s4gchains").exists()
        for n in ("alignments.txt", "LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction"):
            assert (out / n).read_bytes() == open(os.path.join(GOLDEN, "dropin_predictions", n), "rb").read(), (run, n)
    assert sorted(p.name for p in tmp_path.iterdir() if p.is_file()) == ["db.fa", "db.fa.s4gchains"]
    # the full-scan build with the cache: its counting stage takes the totals from the cache header (fastaResidueCount), its
    # alignment stage reads the cached chains: the files of the uncached full-scan run
    if os.path.exists(FULLSCAN):
        out = tmp_path / "fullscan"
        out.mkdir()
        r = subprocess.run([FULLSCAN, "-q", query, "--subst", files + os.sep, "-d", str(db), "--cards", "0",
                            "--out", str(out), "--sub-results"], env=env, cwd=str(out), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        for n in ("alignments.txt", "LACI_ECOLI.SIFTprediction", "PURR_SALTY.SIFTprediction"):
            assert (out / n).read_bytes() == open(os.path.join(GOLDEN, "dropin_fullscan", n), "rb").read(), n
    assert not os.path.exists(os.path.join(files, "query.fasta.s4gchains"))
