"""The pinning recipe (oracle/Makefile target `_ref`, oracle/pin_dump.cpp, oracle/pin_against_swsharp.py): the parts that can run here
do run; the comparison with the reference's real scorer SKIPS VISIBLY while upstream swsharp is an absent submodule
(/root/reference/.gitmodules:1-3, vendor/swsharp empty) -- parity stays "unpinned"  until it does not skip."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

OURS = os.path.join(ROOT, "build", "pin_dump_ours")
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "pin_dump_ref")


def _pin():
    spec = importlib.util.spec_from_file_location("pin_against_swsharp", os.path.join(ROOT, "oracle", "pin_against_swsharp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not os.path.exists(OURS), reason="build/pin_dump_ours not built (make pin-dump)")
def test_this_repos_half_of_the_recipe_reproduces_the_golden_fixture(cfg1):
    """oracle/pin_dump.cpp against this repo's libswsharp, on the oracle stand-in (no GPU here): all 2 x 9 pairs of the reference's
    fixture come back with the golden scores, in e-value order, with paths that start at the reported cells -- so the dumper, which is
    the same source that would be linked against upstream swsharp, asks the questions the comparison needs; and the comparator finds
    nothing between two equal dumps but does find, and attribute to its switch, a change of the e-value convention."""
    pin = _pin()
    tf = os.path.join(GOLDEN, "test_files")
    q, d = os.path.join(tf, "query.fasta"), os.path.join(tf, "sample_protein_database.fa")
    recs = pin.run_dump(OURS, q, d, ("BLOSUM_62", 10, 1), pin.ours_env("oracle"))
    assert len(recs) == 18
    qn, tn = [n for n, _ in cfg1["queries"]], [n for n, _ in cfg1["targets"]]
    for r in recs:
        assert r[2] == cfg1["scores"][qn.index(r[0]), tn.index(r[1])]
        assert set(r[6]) <= set("DLU") and r[6].startswith("D") and r[6].endswith("D") and r[4] >= 0 and r[5] >= 0
    for name in qn:
        vals = [r[3] for r in recs if r[0] == name]
        assert vals == sorted(vals)
    again = pin.run_dump(OURS, q, d, ("BLOSUM_62", 10, 1), pin.ours_env("oracle"))
    assert pin.compare_dumps(recs, again)["equal"]
    blast = pin.run_dump(OURS, q, d, ("BLOSUM_62", 10, 1), pin.ours_env("oracle", {"S4G_EVALUE_CONVENTION": "blast"}))
    cmp_ = pin.compare_dumps(blast, recs)                       # "upstream" played by the other convention
    assert not cmp_["equal"] and cmp_["counts"]["evalues"] == 18 and cmp_["counts"]["scores"] == 0 and cmp_["counts"]["tracebacks"] == 0
    sw, after = pin.explain(blast, q, d, ("BLOSUM_62", 10, 1), "oracle", cmp_)
    assert sw == {"S4G_EVALUE_CONVENTION": "blast"} and after["equal"]


def test_the_recipe_says_unpinned_when_there_is_no_checkout():
    """without oracle/_ref the script exits 77 with the reason on stderr, and `make -C oracle _ref` refuses with the same reason instead
    of building anything from stand-ins"""
    if os.path.exists(REF_DUMP):
        pytest.skip("oracle/_ref is built: see test_against_the_real_swsharp")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "pin_against_swsharp.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "PARITY UNPINNED" in r.stderr
    if os.path.isdir("/root/reference") and not os.listdir("/root/reference/vendor/swsharp"):
        m = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_ref"], capture_output=True, text=True, timeout=120)
        assert m.returncode != 0 and "UNPINNED" in m.stdout and not os.path.exists(REF_DUMP)


@pytest.mark.skipif(not os.path.exists(REF_DUMP),
                    reason="PARITY UNPINNED: upstream swsharp is an absent submodule (/root/reference/.gitmodules:1-3, vendor/swsharp empty, no network); "
                           "with a checkout: make -C oracle _ref SWSHARP_SRC=<dir> && python oracle/pin_against_swsharp.py")
def test_against_the_real_swsharp():
    """scores, e-values, order, tracebacks, alignments.txt and .SIFTprediction files of the reference on upstream swsharp == this repo's"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "pin_against_swsharp.py"), "--ours", "oracle"], capture_output=True, text=True, timeout=7200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
