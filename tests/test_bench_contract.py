"""bench.py contract: one JSON line with the keys the driver reads (GPU), and a stable synthetic workload (CPU)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_synthetic_workload_is_deterministic():
    """the bench workload must not drift between runs, machines or rounds: checksum of a small instance of the generator."""
    from sift4g_amd import synth
    c = synth.make_config(2, n_queries=6, n_targets=400, homologs_per_query=5)
    # digest over the parts produced by integer / exactly-rounded arithmetic only (queries, planted homologs); the background
    # target lengths go through libm's exp/log/cos, whose last ulp may differ between CPUs
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(c["q_codes"]).tobytes()); h.update(np.ascontiguousarray(c["q_off"]).tobytes())
    for _, t in c["planted"]:
        h.update(np.ascontiguousarray(c["t_codes"][int(c["t_off"][t]):int(c["t_off"][t + 1])]).tobytes())
    assert len(c["planted"]) == 30
    ql = np.diff(c["q_off"].astype(np.int64)); tl = np.diff(c["t_off"].astype(np.int64))
    assert ql.min() >= 300 and ql.max() <= 400 and tl.min() >= 30 and tl.max() <= 3000
    assert set(np.unique(c["t_codes"][: int(c["t_off"][-1])]).tolist()) <= set(np.nonzero(synth.AA_FREQUENCY)[0].tolist())
    c2 = synth.make_config(2, n_queries=6, n_targets=400, homologs_per_query=5)
    assert all(np.array_equal(c[k], c2[k]) for k in ("q_codes", "q_off", "t_codes", "t_off"))
    assert h.hexdigest() == "79275571666105ecacc64341f82cf4ff6b415edd831295a851524caccc517dac"


@pytest.mark.gpu
def test_bench_prints_one_contract_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--queries", "16",
                        "--targets", "3000"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "GCUPS" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["achieved"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "GCUPS" and cb["sample"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
