import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the suite (the driver runs `pytest -x`: whatever fails first hides everything behind it).  Row evidence first
# -- tests/test_gpu_rows.py holds one small HIP-vs-oracle test per row of SURVEY.md 8 and per BASELINE config -- then the parity files
# from the kernels outwards, and only then the tests that spawn bench.py ranks or run time-boxed soaks.  Files not listed keep their
# alphabetical place in front of the bench / soak group.
_ORDER = ["test_gpu_rows", "test_gpu_parity", "test_gpu_swsharp", "test_gpu_prefilter", "test_gpu_fullsize", "test_gpu_shard", "test_gpu_multi"]
_LAST = ["test_bench_contract", "test_gpu_soak"]


def _file_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _ORDER:
        return _ORDER.index(name)
    if name in _LAST:
        return len(_ORDER) + 1 + _LAST.index(name)
    return len(_ORDER)


def pytest_collection_modifyitems(config, items):
    """(1) the order above (a stable sort: the order inside a file is the file's own);  (2) a test that hangs (a kernel that never
    ends, a deadlocked rendezvous) must fail on its own instead of holding the GPU box until the driver's limit: with pytest-timeout
    installed every test gets 15 minutes unless it asked for something else (the longest, the full-size configs, take about one)."""
    items.sort(key=_file_rank)
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))


# The driver keeps the TAIL of the log, and `-q` prints dots: one line at the very end says, row by row of SURVEY.md 8 and config by config of
# BASELINE.json, what tests/test_gpu_rows.py found -- also when a later, heavier test failed and `-x` stopped the run behind it.
_ROW_OUTCOMES = {}


def pytest_runtest_logreport(report):
    if "test_gpu_rows.py::" not in report.nodeid:
        return
    name = report.nodeid.split("::", 1)[1]
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        _ROW_OUTCOMES[name] = report.outcome if report.when == "call" else f"{report.outcome} in setup"


def pytest_terminal_summary(terminalreporter):
    if not _ROW_OUTCOMES:
        return
    rows = {}
    for name, outcome in _ROW_OUTCOMES.items():
        parts = name[len("test_"):].split("_")
        tags = [parts[0]] + ([parts[1]] if len(parts) > 1 and parts[1].startswith("cfg") else [])      # test_a2_cfg0_... covers two
        for tag in tags:
            key = {"dagger": "cli"}.get(tag, tag)
            worst = rows.get(key)
            rows[key] = outcome if worst in (None, "passed") else worst
    ok = sum(1 for v in rows.values() if v == "passed")
    terminalreporter.write_line(f"SURVEY 8 rows / BASELINE configs (tests/test_gpu_rows.py, HIP path == oracle): {ok} of {len(rows)} passed: "
                                + " ".join(f"{k}={v}" for k, v in rows.items()))


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()  # builds oracle/libsw_oracle.so on first use
    return oracle


@pytest.fixture(scope="session")
def blosum62(oracle_mod):
    return oracle_mod.blosum62()


@pytest.fixture(scope="session")
def micro():
    with open(os.path.join(GOLDEN, "micro.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def cfg1():
    """BASELINE.json configs[0]: the reference's own test_files (inputs) + oracle-generated scores."""
    from sift4g_amd.seqio import read_fasta
    with open(os.path.join(GOLDEN, "cfg1_scores.json")) as fh:
        gold = json.load(fh)
    queries = read_fasta(os.path.join(GOLDEN, "test_files", "query.fasta"))
    targets = read_fasta(os.path.join(GOLDEN, "test_files", "sample_protein_database.fa"))
    return dict(queries=queries, targets=targets, scores=np.array(gold["scores"], dtype=np.int32), gold=gold)


def rand_seqs(rng, n, lo, hi, alphabet=20):
    """n random code arrays with lengths in [lo, hi]; alphabet=20 -> standard residues, 26 -> every letter code."""
    std = np.array([0, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 15, 16, 17, 18, 19, 21, 22, 24], dtype=np.uint8)
    out = []
    for _ in range(n):
        l = int(rng.integers(lo, hi + 1))
        if alphabet == 26:
            out.append(rng.integers(0, 26, size=l).astype(np.uint8))
        else:
            out.append(std[rng.integers(0, 20, size=l)])
    return out
